// Yaz0.cs -- drop-in for AuroraLib.Compression.Formats.Nintendo.Yaz0 (src/AuroraLib.Compression.Nintendo/Nintendo/Yaz0.cs).
using AuroraLib.Compression.Interfaces;
using AuroraLib.Core.Format;
using AuroraLib.Core.Format.Identifier;
using AuroraLib.Core.IO;
using System;
using System.Buffers.Binary;
using System.IO;
using Managed = AuroraLib.Compression.Formats.Nintendo;

namespace AuroraLib.Compression.Amd.Nintendo
{
    public class Yaz0 : ICompressionAlgorithm, IEndianDependentFormat, IProvidesDecompressedSize
    {
        private static readonly Identifier32 _identifier = new Identifier32("Yaz0".AsSpan());
        private static readonly IFormatInfo _info = new FormatInfo<Yaz0>("Nintendo Yaz0 (MI355X)", new MediaType(MIMEType.Application, "x-nintendo-yaz0"), string.Empty, _identifier);

        /// <inheritdoc/>
        public virtual IFormatInfo Info => _info;
        public virtual IIdentifier Identifier => _identifier;

        /// <inheritdoc/>
        public Endian FormatByteOrder { get; set; } = Endian.Big;       // Yaz0.cs:30
        public uint MemoryAlignment { get; set; }                       // Yaz0.cs:39

        /// <inheritdoc/>
        public virtual bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => stream.Position + 0x10 < stream.Length && stream.Peek(s => s.Match(Identifier.AsSpan()));   // Yaz0.cs:46-47

        /// <inheritdoc/>
        public uint GetDecompressedSize(Stream source)
            => source.Peek(s => { s.MatchThrow(Identifier.AsSpan()); return s.ReadUInt32(FormatByteOrder); });

        /// <inheritdoc/>
        public virtual void Decompress(Stream source, Stream destination)   // Yaz0.cs:58-79, including the byte-order retry
        {
            source.MatchThrow(Identifier.AsSpan());
            uint size = source.ReadUInt32(FormatByteOrder);
            MemoryAlignment = source.ReadUInt32(FormatByteOrder);
            _ = source.ReadUInt32(FormatByteOrder);
            long sourceStart = source.Position, destinationStart = destination.Position;
            try
            {
                DecompressHeaderless(source, destination, size);
            }
            catch (Exception)
            {
                source.Seek(sourceStart, SeekOrigin.Begin);
                destination.Seek(destinationStart, SeekOrigin.Begin);
                size = BinaryPrimitives.ReverseEndianness(size);
                MemoryAlignment = BinaryPrimitives.ReverseEndianness(MemoryAlignment);
                DecompressHeaderless(source, destination, size);
            }
        }

        /// <summary>Yaz0.DecompressHeaderless (Yaz0.cs:91-92 = Yay0.cs:110-144 with all three cursors on one stream).</summary>
        public static unsafe void DecompressHeaderless(Stream source, Stream destination, uint decomLength)
        {
            if (!AmdBody.UseGpuBigStream(decomLength)) { Managed.Yaz0.DecompressHeaderless(source, destination, decomLength); return; }
            AmdBody.Decode(AlzFormat.Yaz0, null, source, destination, decomLength, 0, 0, decomLength + 273, true);
        }

        /// <inheritdoc/>
        public virtual void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)   // Yaz0.cs:82-89
        {
            destination.Write(Identifier.AsSpan());
            destination.Write(source.Length, FormatByteOrder);
            destination.Write(MemoryAlignment, FormatByteOrder);
            destination.Write(0);
            CompressHeaderless(source, destination, settings);
        }

        /// <summary>Yaz0.CompressHeaderless (Yaz0.cs:94-127).</summary>
        public static unsafe void CompressHeaderless(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)
        {
            if (!AmdBody.UseGpuForCompress(AlzFormat.Yaz0, source.Length, settings)) { Managed.Yaz0.CompressHeaderless(source, destination, settings); return; }
            AmdBody.Encode(AlzFormat.Yaz0, null, source, destination, settings, 0);
        }
    }
}

// MIO0.cs -- drop-in for AuroraLib.Compression.Formats.Nintendo.MIO0 (src/AuroraLib.Compression.Nintendo/Nintendo/MIO0.cs).
using AuroraLib.Compression.Interfaces;
using AuroraLib.Core.Format;
using AuroraLib.Core.Format.Identifier;
using AuroraLib.Core.IO;
using System;
using System.IO;
using Managed = AuroraLib.Compression.Formats.Nintendo;

namespace AuroraLib.Compression.Amd.Nintendo
{
    public sealed class MIO0 : ICompressionAlgorithm, IEndianDependentFormat, IProvidesDecompressedSize
    {
        private static readonly Identifier32 _identifier = new Identifier32("MIO0".AsSpan());
        private static readonly IFormatInfo _info = new FormatInfo<MIO0>("Nintendo MIO0 (MI355X)", new MediaType(MIMEType.Application, "x-nintendo-mio0"), string.Empty, _identifier);

        /// <inheritdoc/>
        public IFormatInfo Info => _info;

        /// <inheritdoc/>
        public Endian FormatByteOrder { get; set; } = Endian.Big;       // MIO0.cs:30

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => Managed.MIO0.IsMatchStatic(stream, fileNameAndExtension);

        /// <inheritdoc/>
        public uint GetDecompressedSize(Stream source)
            => source.Peek(s => { s.MatchThrow(_identifier); return s.ReadUInt32(s.DetectByteOrder<uint>(3)); });   // MIO0.cs:42-49

        /// <inheritdoc/>
        public void Decompress(Stream source, Stream destination)        // MIO0.cs:51-61
        {
            const int flagDataStart = 0x10;
            uint startPosition = (uint)source.Position;
            source.MatchThrow(_identifier);
            Endian endian = source.DetectByteOrder<uint>(3);
            uint size = source.ReadUInt32(endian);
            uint compressedDataPointer = source.ReadUInt32(endian) + startPosition;
            uint uncompressedDataPointer = source.ReadUInt32(endian) + startPosition;
            DecompressHeaderless(source, destination, size, (int)compressedDataPointer - flagDataStart, (int)uncompressedDataPointer - flagDataStart);
        }

        /// <summary>MIO0.DecompressHeaderless (MIO0.cs:82-149): flags from the current position, tokens and literals at the two
        /// pointers (relative to the first flag byte); source.Position ends behind the last section byte read (:92-93).</summary>
        public static unsafe void DecompressHeaderless(Stream source, Stream destination, uint decomLength, int compressedDataPointer, int uncompressedDataPointer)
        {
            if (!AmdBody.UseGpuBigStream(decomLength)) { Managed.MIO0.DecompressHeaderless(source, destination, decomLength, compressedDataPointer, uncompressedDataPointer); return; }
            AmdBody.Decode(AlzFormat.MIO0, null, source, destination, decomLength, (uint)compressedDataPointer, (uint)uncompressedDataPointer, decomLength + 18, true);
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)   // MIO0.cs:63-80
        {
            if (!AmdBody.UseGpuForCompress(AlzFormat.MIO0, source.Length, settings)) { var m = new Managed.MIO0 { FormatByteOrder = FormatByteOrder }; m.Compress(source, destination, settings); return; }
            using (MemoryStream body = new MemoryStream())
            {
                // the native encoder writes flags | tokens | literals back to back and reports where the sections start
                AlzEncodeAux aux;
                unsafe { aux = AmdBody.Encode(AlzFormat.MIO0, null, source, body, settings, 0); }
                uint startPosition = (uint)destination.Position;
                destination.Write(_identifier);
                destination.Write(source.Length, FormatByteOrder);
                destination.Write((uint)(0x10 + aux.Aux0 - startPosition), FormatByteOrder);
                destination.Write((uint)(0x10 + aux.Aux1 - startPosition), FormatByteOrder);
                body.WriteTo(destination);
            }
        }
    }
}

// LZ10.cs -- drop-in for AuroraLib.Compression.Formats.Nintendo.LZ10 (src/AuroraLib.Compression.Nintendo/Nintendo/LZ10.cs):
// same interfaces, same header code, the DecompressHeaderless / CompressHeaderless bodies behind the C ABI.
using AuroraLib.Compression.Interfaces;
using AuroraLib.Core.Exceptions;
using AuroraLib.Core.Format;
using AuroraLib.Core.IO;
using System;
using System.IO;
using Managed = AuroraLib.Compression.Formats.Nintendo;

namespace AuroraLib.Compression.Amd.Nintendo
{
    public sealed class LZ10 : ICompressionAlgorithm, IProvidesDecompressedSize, Managed.IGbaRamMode
    {
        private const byte Identifier = 0x10;
        private static readonly IFormatInfo _info = new FormatInfo<LZ10>("Nintendo LZ10 (MI355X)", new MediaType(MIMEType.Application, "x-nintendo-lz10"), ".lz");

        /// <inheritdoc/>
        public IFormatInfo Info => _info;

        /// <inheritdoc/>
        public bool GbaVramCompatibilityMode { get; set; } = true;      // LZ10.cs:33

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => Managed.LZ10.IsMatchStatic(stream, fileNameAndExtension);  // LZ10.cs:36-41: the reference's own check

        /// <inheritdoc/>
        public uint GetDecompressedSize(Stream source) => source.Peek(ReadHeader);

        private static uint ReadHeader(Stream source)                    // LZ10.cs:47-57
        {
            byte identifier = source.ReadUInt8();
            if (identifier != Identifier)
                throw new InvalidIdentifierException(identifier.ToString("X"), Identifier.ToString("X"));
            uint size = source.ReadUInt24();
            if (size == 0)
                size = source.ReadUInt32();
            return size;
        }

        /// <inheritdoc/>
        public void Decompress(Stream source, Stream destination)
            => DecompressHeaderless(source, destination, ReadHeader(source));

        /// <summary>LZ10.DecompressHeaderless (LZ10.cs:82-111): overshoot of the declared size is the error ('>', :107).</summary>
        public static unsafe void DecompressHeaderless(Stream source, Stream destination, uint decomLength)
        {
            if (!AmdBody.UseGpuBigStream(decomLength)) { Managed.LZ10.DecompressHeaderless(source, destination, decomLength); return; }
            AmdBody.Decode(AlzFormat.LZ10, null, source, destination, decomLength, 0, 0, decomLength + 18, true);
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)
        {
            if (source.Length <= 0xFFFFFF)                              // LZ10.cs:67-77
                destination.Write(Identifier | (source.Length << 8));
            else
            {
                destination.Write(Identifier | 0);
                destination.Write(source.Length);
            }
            CompressHeaderless(source, destination, settings, GbaVramCompatibilityMode);
        }

        /// <summary>LZ10.CompressHeaderless (LZ10.cs:113-137) + LzChainMatchFinder: bit-identical output at every quality.</summary>
        public static unsafe void CompressHeaderless(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default, bool gbaVramCompatibilityMode = true)
        {
            if (!AmdBody.UseGpuForCompress(AlzFormat.LZ10, source.Length, settings)) { Managed.LZ10.CompressHeaderless(source, destination, settings, gbaVramCompatibilityMode); return; }
            AmdBody.Encode(AlzFormat.LZ10, null, source, destination, settings, gbaVramCompatibilityMode ? 2 : 1);
        }
    }
}

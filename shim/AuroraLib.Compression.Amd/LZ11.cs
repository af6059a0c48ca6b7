// LZ11.cs -- drop-in for AuroraLib.Compression.Formats.Nintendo.LZ11 (src/AuroraLib.Compression.Nintendo/Nintendo/LZ11.cs).
using AuroraLib.Compression.Interfaces;
using AuroraLib.Core.Exceptions;
using AuroraLib.Core.Format;
using AuroraLib.Core.IO;
using System;
using System.IO;
using Managed = AuroraLib.Compression.Formats.Nintendo;

namespace AuroraLib.Compression.Amd.Nintendo
{
    public sealed class LZ11 : ICompressionAlgorithm, IProvidesDecompressedSize, Managed.IGbaRamMode
    {
        private const byte Identifier = 0x11;
        private static readonly IFormatInfo _info = new FormatInfo<LZ11>("Nintendo LZ11 (MI355X)", new MediaType(MIMEType.Application, "x-nintendo-lz11"), ".lz");

        /// <inheritdoc/>
        public IFormatInfo Info => _info;

        /// <inheritdoc/>
        public bool GbaVramCompatibilityMode { get; set; } = false;     // LZ11.cs:29

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => Managed.LZ11.IsMatchStatic(stream, fileNameAndExtension);

        /// <inheritdoc/>
        public uint GetDecompressedSize(Stream source) => source.Peek(ReadHeader);

        private static uint ReadHeader(Stream source)                    // LZ11.cs:43-53
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

        /// <summary>LZ11.DecompressHeaderless (LZ11.cs:83-133); the longest token is 65 808 bytes.</summary>
        public static unsafe void DecompressHeaderless(Stream source, Stream destination, uint decomLength)
        {
            if (!AmdBody.UseGpuBigStream(decomLength)) { Managed.LZ11.DecompressHeaderless(source, destination, decomLength); return; }
            AmdBody.Decode(AlzFormat.LZ11, null, source, destination, decomLength, 0, 0, decomLength + 65808, true);
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)
        {
            if (source.Length <= 0xFFFFFF)                              // LZ11.cs:66-76
                destination.Write(Identifier | (source.Length << 8));
            else
            {
                destination.Write(Identifier | 0);
                destination.Write(source.Length);
            }
            CompressHeaderless(source, destination, settings, GbaVramCompatibilityMode);
        }

        /// <summary>LZ11.CompressHeaderless (LZ11.cs:135-171).</summary>
        public static unsafe void CompressHeaderless(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default, bool gbaVramCompatibilityMode = false)
        {
            if (!AmdBody.UseGpuForCompress(AlzFormat.LZ11, source.Length, settings)) { Managed.LZ11.CompressHeaderless(source, destination, settings, gbaVramCompatibilityMode); return; }
            AmdBody.Encode(AlzFormat.LZ11, null, source, destination, settings, gbaVramCompatibilityMode ? 2 : 1);
        }
    }
}

// LZO.cs -- drop-in for AuroraLib.Compression.Formats.Common.LZO (src/AuroraLib.Compression/Formats/Common/LZO.cs).
using AuroraLib.Compression.Interfaces;
using AuroraLib.Core.Format;
using System;
using System.IO;
using Managed = AuroraLib.Compression.Formats.Common;

namespace AuroraLib.Compression.Amd.Common
{
    public sealed class LZO : ICompressionAlgorithm
    {
        private static readonly IFormatInfo _info = new FormatInfo<LZO>("Lempel-Ziv-Oberhumer (MI355X)", new MediaType(MIMEType.Application, "x-lzo"), ".lzo");

        /// <inheritdoc/>
        public IFormatInfo Info => _info;

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => Managed.LZO.IsMatchStatic(stream, fileNameAndExtension);   // LZO.cs:31-40 (file extension only)

        /// <inheritdoc/>
        public void Decompress(Stream source, Stream destination) => DecompressHeaderless(source, destination);

        /// <summary>LZO.DecompressHeaderless (LZO.cs:49-139), including its first-byte quirk (:59-64); no size field: runs to the
        /// end marker, the destination capacity doubles while the body reports OUTPUT_CAPACITY.</summary>
        public static unsafe void DecompressHeaderless(Stream source, Stream destination)
        {
            long rest = source.Length - source.Position;
            if (!AmdBody.UseGpuBigStream((uint)Math.Min(uint.MaxValue, rest * 4))) { Managed.LZO.DecompressHeaderless(source, destination); return; }
            AmdBody.Decode(AlzFormat.LZO, null, source, destination, 0, 0, 0, (uint)Math.Min(0x7FFF0000L, Math.Max(4096L, rest * 8)), false);
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)
            => CompressHeaderless(source, destination, settings);

        /// <summary>LZO.CompressHeaderless (LZO.cs:141-250), quirks included (the literal-run padding of :167-172).</summary>
        public static unsafe void CompressHeaderless(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)
        {
            if (!AmdBody.UseGpuForCompress(AlzFormat.LZO, source.Length, settings)) { Managed.LZO.CompressHeaderless(source, destination, settings); return; }
            AmdBody.Encode(AlzFormat.LZO, null, source, destination, settings, 0);
        }
    }
}

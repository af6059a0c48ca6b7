// PRS.cs -- drop-in for AuroraLib.Compression.Formats.Sega.PRS (src/AuroraLib.Compression.Sega/Sega/PRS.cs).
using AuroraLib.Compression.Interfaces;
using AuroraLib.Compression.IO;
using AuroraLib.Core.Format;
using AuroraLib.Core.IO;
using System;
using System.IO;
using Managed = AuroraLib.Compression.Formats.Sega;

namespace AuroraLib.Compression.Amd.Sega
{
    public sealed class PRS : ICompressionAlgorithm, IEndianDependentFormat
    {
        private static readonly IFormatInfo _info = new FormatInfo<PRS>("SEGA PRS (MI355X)", new MediaType(MIMEType.Application, "x-sega-prs"), ".prs");

        /// <inheritdoc/>
        public IFormatInfo Info => _info;

        /// <inheritdoc/>
        public Endian FormatByteOrder { get; set; } = Endian.Little;     // PRS.cs:24

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => Managed.PRS.IsMatchStatic(stream, fileNameAndExtension);   // PRS.cs:31-32 (byte-order detection on the last bytes)

        /// <inheritdoc/>
        public void Decompress(Stream source, Stream destination) => DecompressHeaderless(source, destination);

        /// <summary>PRS.DecompressHeaderless(Stream, Stream) (PRS.cs:42-57): detected byte order first, the other one when that throws.</summary>
        public static void DecompressHeaderless(Stream source, Stream destination)
        {
            if (!AmdBody.UseGpu((uint)Math.Min(uint.MaxValue, (source.Length - source.Position) * 4))) { Managed.PRS.DecompressHeaderless(source, destination); return; }
            Endian detected = GetByteOrder(source) == Endian.Big ? Endian.Big : Endian.Little;
            long sourcePos = source.Position, destinationPos = destination.Position;
            try
            {
                DecompressHeaderless(source, destination, detected);
            }
            catch (Exception)
            {
                source.Seek(sourcePos, SeekOrigin.Begin);
                destination.Seek(destinationPos, SeekOrigin.Begin);
                DecompressHeaderless(source, destination, detected == Endian.Big ? Endian.Little : Endian.Big);
            }
        }

        /// <summary>PRS.DecompressHeaderless(Stream, Stream, Endian) (PRS.cs:59-102): no size anywhere -- the stream runs to its
        /// zero word; the destination capacity starts at 8x the input and doubles while the body reports OUTPUT_CAPACITY.</summary>
        public static unsafe void DecompressHeaderless(Stream source, Stream destination, Endian order)
        {
            long rest = source.Length - source.Position;
            uint guess = (uint)Math.Min(0x7FFF0000L, Math.Max(4096L, rest * 8));
            AmdBody.Decode(order == Endian.Big ? AlzFormat.PrsBE : AlzFormat.PrsLE, null, source, destination, 0, 0, 0, guess, false);
        }

        // PRS.cs:161-218 (private there): which bit / byte order do the first tokens make sense in?
        private static Endian? GetByteOrder(Stream stream)
        {
            byte flag = stream.PeekByte();
            if (flag > 12 && (flag & 0x1) == 1 && ValidateByteOrder(stream, Endian.Little))
                return Endian.Little;
            if ((flag & 128) == 128 && ValidateByteOrder(stream, Endian.Big))
                return Endian.Big;
            return null;
        }

        private static bool ValidateByteOrder(Stream stream, Endian order)
        {
            int i = 3, produced = 0;
            long startPos = stream.Position;
            FlagReader flag = new FlagReader(stream, order);
            try
            {
                while (stream.Position < stream.Length)
                {
                    if (flag.Readbit()) { stream.Position++; produced++; continue; }
                    int distance, length;
                    if (flag.Readbit())
                    {
                        distance = stream.ReadUInt16(order);
                        if (distance == 0) return true;
                        length = distance & 7;
                        distance = 0x2000 - (distance >> 3);
                        length = length == 0 ? stream.ReadUInt8() + 1 : length + 2;
                    }
                    else
                    {
                        length = flag.ReadInt(2, true) + 2;
                        distance = 0x100 - stream.ReadUInt8();
                    }
                    if (distance > produced) return false;
                    if (i == 0) return true;
                    i--;
                    produced += length;
                }
                return false;
            }
            finally { stream.Position = startPos; }
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)
            => CompressHeaderless(source, destination, FormatByteOrder, settings);

        /// <summary>PRS.CompressHeaderless (PRS.cs:104-159).</summary>
        public static unsafe void CompressHeaderless(ReadOnlySpan<byte> source, Stream destination, Endian order = Endian.Little, CompressionSettings settings = default)
        {
            if (!AmdContext.Available) { Managed.PRS.CompressHeaderless(source, destination, order, settings); return; }
            AmdBody.Encode(order == Endian.Big ? AlzFormat.PrsBE : AlzFormat.PrsLE, null, source, destination, settings, 0);
        }
    }
}

// PRS.cs -- drop-in for AuroraLib.Compression.Formats.Sega.PRS (src/AuroraLib.Compression.Sega/Sega/PRS.cs).
using AuroraLib.Compression.Interfaces;
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
        public Endian FormatByteOrder { get; set; } = Endian.Big;        // PRS.cs:23 (the static CompressHeaderless defaults to Little, :104)

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => Managed.PRS.IsMatchStatic(stream, fileNameAndExtension);   // PRS.cs:31-32 (byte-order detection on the first tokens)

        /// <inheritdoc/>
        public void Decompress(Stream source, Stream destination) => DecompressHeaderless(source, destination);

        /// <summary>PRS.DecompressHeaderless(Stream, Stream) (PRS.cs:42-57): the byte order the stream opens plausibly in first, the
        /// other one when that attempt throws.  The rest of the source is read ONCE; both attempts decode the same buffer.</summary>
        public static void DecompressHeaderless(Stream source, Stream destination)
        {
            if (!AmdBody.UseGpuBigStream((uint)Math.Min(uint.MaxValue, (source.Length - source.Position) * 4))) { Managed.PRS.DecompressHeaderless(source, destination); return; }
            long destinationPos = destination.Position;
            byte[] body = AmdBody.RentRest(source, out int length);
            try
            {
                // PRS.cs:44, :160-169: little-endian is asked first, big-endian second, neither means little-endian
                Endian first = !PrsOpening.LooksLike(body, length, Endian.Little) && PrsOpening.LooksLike(body, length, Endian.Big) ? Endian.Big : Endian.Little;
                try { Decode(body, length, source, destination, first); }
                catch (Exception)
                {
                    destination.Seek(destinationPos, SeekOrigin.Begin);
                    Decode(body, length, source, destination, first == Endian.Big ? Endian.Little : Endian.Big);
                }
            }
            finally { System.Buffers.ArrayPool<byte>.Shared.Return(body); }
        }

        /// <summary>PRS.DecompressHeaderless(Stream, Stream, Endian) (PRS.cs:59-102): no size anywhere -- the stream runs to its
        /// zero word; the destination capacity starts at 8x the input and doubles while the body reports OUTPUT_CAPACITY.</summary>
        public static void DecompressHeaderless(Stream source, Stream destination, Endian order)
        {
            byte[] body = AmdBody.RentRest(source, out int length);
            try { Decode(body, length, source, destination, order); }
            finally { System.Buffers.ArrayPool<byte>.Shared.Return(body); }
        }

        private static unsafe void Decode(byte[] body, int length, Stream source, Stream destination, Endian order)
        {
            uint guess = (uint)Math.Min(0x7FFF0000L, Math.Max(4096L, (long)length * 8));
            AmdBody.DecodeRented(order == Endian.Big ? AlzFormat.PrsBE : AlzFormat.PrsLE, null, body, length, source, destination, 0, 0, 0, guess, false);
        }

        /// <summary>
        /// The reference picks its first attempt by looking at how the stream opens (PRS.cs:161-218, private there): a first flag
        /// byte that starts with a literal bit in the order in question, followed by tokens whose first four matches all reach
        /// back no further than the bytes produced so far (or by the zero word).  Restated over the rented buffer: a bit cursor
        /// over bytes, no Stream and no FlagReader.
        /// </summary>
        private static class PrsOpening
        {
            internal static bool LooksLike(byte[] d, int n, Endian order)
            {
                if (n == 0) return false;
                bool big = order == Endian.Big;
                if (big ? (d[0] & 0x80) == 0 : !(d[0] > 12 && (d[0] & 1) == 1)) return false;     // PRS.cs:164-169
                int at = 0, held = 0, left = 0, written = 0, matchesToSee = 4;
                while (at < n)
                {
                    int kind = 0;                                         // control bits of one token: "1", "01", "00hl"
                    int extra = 0;
                    for (int need = 1; need > 0; need--)
                    {
                        if (left == 0) { if (at >= n) return false; held = d[at++]; left = 8; }
                        int bit = big ? (held >> (left - 1)) & 1 : (held >> (8 - left)) & 1;
                        left--;
                        if (kind == 0) { if (bit == 1) kind = 1; else { kind = 2; need = 2; } }
                        else if (kind == 2) { if (bit == 1) kind = 3; else { kind = 4; need = 3; } }
                        else extra = (extra << 1) | bit;
                    }
                    if (kind == 1) { at++; written++; continue; }
                    int reach, span;
                    if (kind == 3)
                    {
                        if (at + 2 > n) return false;
                        int word = big ? (d[at] << 8) | d[at + 1] : (d[at + 1] << 8) | d[at];
                        at += 2;
                        if (word == 0) return true;
                        reach = 0x2000 - (word >> 3);
                        if ((word & 7) != 0) span = (word & 7) + 2;
                        else { if (at >= n) return false; span = d[at++] + 1; }
                    }
                    else
                    {
                        if (at >= n) return false;
                        span = extra + 2;
                        reach = 0x100 - d[at++];
                    }
                    if (reach > written) return false;
                    if (--matchesToSee == 0) return true;
                    written += span;
                }
                return false;
            }
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)
            => CompressHeaderless(source, destination, FormatByteOrder, settings);

        /// <summary>PRS.CompressHeaderless (PRS.cs:104-159).</summary>
        public static unsafe void CompressHeaderless(ReadOnlySpan<byte> source, Stream destination, Endian order = Endian.Little, CompressionSettings settings = default)
        {
            if (!AmdBody.UseGpuForCompress(order == Endian.Big ? AlzFormat.PrsBE : AlzFormat.PrsLE, source.Length, settings)) { Managed.PRS.CompressHeaderless(source, destination, order, settings); return; }
            AmdBody.Encode(order == Endian.Big ? AlzFormat.PrsBE : AlzFormat.PrsLE, null, source, destination, settings, 0);
        }
    }
}

// AmdBody.cs -- what every format class of this assembly shares: move the rest of a Stream through a native body decode /
// encode and map the per-stream status back to the exception the managed body would have thrown.
using AuroraLib.Compression.Exceptions;
using System;
using System.Buffers;
using System.IO;

namespace AuroraLib.Compression.Amd
{
    internal static unsafe class AmdBody
    {
        /// <summary>How a body treats its declared size (resolve_status of the kernels mirrors the same rules).</summary>
        internal enum SizeRule { None, OvershootOnly /* LZ10.cs:107 '>' */, MustMatch /* LZSS.cs:126 '!=' */ }

        /// <summary>Reads what is left of <paramref name="source"/> (netstandard2.0: no Stream.ReadExactly).</summary>
        internal static byte[] RentRest(Stream source, out int length)
        {
            long rest = source.Length - source.Position;
            if (rest > int.MaxValue - 64) throw new NotSupportedException("streams above 2 GiB are not supported by the native body");
            length = (int)rest;
            byte[] buf = ArrayPool<byte>.Shared.Rent(length + 16);
            int done = 0;
            while (done < length)
            {
                int n = source.Read(buf, done, length - done);
                if (n <= 0) throw new EndOfStreamException();
                done += n;
            }
            return buf;
        }

        /// <summary>
        /// The replacement of a static DecompressHeaderless(Stream source, Stream destination, uint decomLength): the body runs in
        /// alz_decode, the output is written to <paramref name="destination"/> (partial output too, as the managed bodies leave
        /// it), <c>source.Position</c> ends just past the consumed bytes (Yay0.cs:89-90, MIO0.cs:92-93, LZSS.cs:68) and the
        /// status becomes the managed exception.  <paramref name="capacity"/>: bytes the body may produce (declared size plus the
        /// longest token for bodies that may overshoot it; a guess that is doubled on OUTPUT_CAPACITY for bodies without a size).
        /// </summary>
        internal static void Decode(AlzFormat format, AlzLzProperties* props, Stream source, Stream destination, uint decomLength,
                                    uint aux0, uint aux1, uint capacity, bool hasSize)
        {
            byte[] src = RentRest(source, out int srcLen);
            try { DecodeRented(format, props, src, srcLen, source, destination, decomLength, aux0, aux1, capacity, hasSize); }
            finally { ArrayPool<byte>.Shared.Return(src); }
        }

        /// <summary>The same over a buffer that already holds the rest of <paramref name="source"/> (whose position is at its end):
        /// formats that may decode one input twice (PRS: the other byte order, PRS.cs:42-57) rent it once.</summary>
        internal static void DecodeRented(AlzFormat format, AlzLzProperties* props, byte[] src, int srcLen, Stream source, Stream destination,
                                          uint decomLength, uint aux0, uint aux1, uint capacity, bool hasSize)
        {
            {
                for (;;)
                {
                    byte[] dst = ArrayPool<byte>.Shared.Rent((int)Math.Min(capacity, int.MaxValue - 64));
                    try
                    {
                        AlzResult r;
                        lock (AmdContext.Lock)
                            fixed (byte* ps = src, pd = dst)
                                AmdContext.Check(Native.alz_decode(AmdContext.Handle, (uint)format, props, ps, (uint)srcLen, decomLength, aux0, aux1,
                                                                   pd, (uint)Math.Min(capacity, (uint)dst.Length), &r));
                        if (!hasSize && r.Status == (int)AlzStatus.OutputCapacity && capacity < 0x7FFF0000u)
                        {
                            capacity = capacity < 0x3FFF0000u ? capacity * 2 : 0x7FFF0000u;    // a managed destination Stream simply grows
                            continue;
                        }
                        destination.Write(dst, 0, (int)r.DstLen);
                        if (source.CanSeek && (r.Status == (int)AlzStatus.Ok || r.Status == (int)AlzStatus.OutputSizeMismatch))
                            source.Position -= srcLen - (int)r.SrcUsed;
                        ThrowForStatus((AlzStatus)r.Status, decomLength, r.DstLen);
                        return;
                    }
                    finally { ArrayPool<byte>.Shared.Return(dst); }
                }
            }
        }

        /// <summary>INTEGRATION.md section 3: per-stream status -> the exception of the managed body.</summary>
        internal static void ThrowForStatus(AlzStatus status, long expected, long actual)
        {
            switch (status)
            {
                case AlzStatus.Ok: return;
                case AlzStatus.InputTruncated: throw new EndOfStreamException();                    // PRS.cs:101, LZO.cs:138
                case AlzStatus.OutputSizeMismatch: throw new DecompressedSizeException(expected, actual);   // LZ10.cs:107-110, LZSS.cs:126-129
                case AlzStatus.OutputCapacity: throw new NotSupportedException("destination too small");  // CompressionExtension.cs:43-50
                default: throw new InvalidDataException("token outside the format (reference-undefined input)");
            }
        }

        /// <summary>
        /// The replacement of a static CompressHeaderless(ReadOnlySpan&lt;byte&gt;, Stream, CompressionSettings): alz_encode_batch
        /// with one entry.  Returns the section offsets of Yay0 / MIO0 (flags | tokens | literals are written back to back).
        /// </summary>
        internal static AlzEncodeAux Encode(AlzFormat format, AlzLzProperties* props, ReadOnlySpan<byte> source, Stream destination,
                                            CompressionSettings settings, int minDistance)
        {
            int cap = source.Length + source.Length / 4 + 64;                  // worst case of every body on the path
            byte[] dst = ArrayPool<byte>.Shared.Rent(cap);
            try
            {
                AlzStream st = new AlzStream { SrcOff = 0, DstOff = 0, SrcLen = (uint)source.Length, DstCap = (uint)cap, Format = (uint)format };
                AlzSettings s = new AlzSettings { Quality = settings.Quality, MaxWindowBits = settings.MaxWindowBits,
                                                  Strategy = (int)settings.Strategy, MinDistance = minDistance };
                AlzResult r; AlzEncodeAux aux;
                lock (AmdContext.Lock)
                    fixed (byte* ps = source, pd = dst)
                        AmdContext.Check(Native.alz_encode_batch(AmdContext.Handle, props, &s, 1, ps, (UIntPtr)(uint)source.Length, &st,
                                                                 pd, (UIntPtr)(uint)cap, &r, &aux));
                if (r.Status != (int)AlzStatus.Ok) throw new InvalidOperationException("native encoder status " + r.Status);
                destination.Write(dst, 0, (int)r.DstLen);
                return aux;
            }
            finally { ArrayPool<byte>.Shared.Return(dst); }
        }

        /// <summary>True when a single stream of this size should run on the GPU rather than on the managed body.</summary>
        internal static bool UseGpu(uint decomLength) => AmdContext.Available && decomLength >= AmdContext.SingleStreamThreshold;

        /// <summary>The same for ONE LZSS / LZ10 / LZ11 / Yaz0 / Yay0 / MIO0 / PRS / LZO stream, which the native library decodes on the whole GPU
        /// (alz_ctx_big_stream).</summary>
        internal static bool UseGpuBigStream(uint decomLength)
            => AmdContext.Available && decomLength >= Math.Min(AmdContext.SingleStreamThreshold, AmdContext.BigStreamThreshold);

        /// <summary>True when ONE buffer should be compressed by the native encoder.  A caller's <c>MaxWindowBits</c> only ever WIDENS
        /// the managed finder (LzChainMatchFinder.cs:69-73): within the format's own window it changes nothing and the native
        /// encoder takes it; beyond it (where the managed finder returns distances the format cannot store) the call stays managed,
        /// except for FastLZ, whose level-2 switch it is (FastLZ.cs:163-170).</summary>
        internal static bool UseGpuForCompress(AlzFormat format, int sourceLength, CompressionSettings settings, LzProperties? lz = null)
            => MaxWindowBitsOnGpu(format, settings.MaxWindowBits, lz)
               && ((uint)sourceLength >= AmdContext.SingleStreamCompressThreshold || BigStreamCompress(format, sourceLength, settings, lz))
               && AmdContext.Available;

        /// <summary>ONE buffer the native encoder runs on the whole GPU (csrc/alz_encode_big.h; alz_encode_big_eligible): a format of that
        /// path, at least <see cref="AmdContext.BigStreamCompressThreshold"/> bytes, distances within 16-bit links, and a quality at which the path
        /// beats the managed encoder (<see cref="AmdContext.BigStreamCompressMaxQuality"/>).</summary>
        internal static bool BigStreamCompress(AlzFormat format, int sourceLength, CompressionSettings settings, LzProperties? lz = null)
        {
            switch (format)
            {
                case AlzFormat.LZSS: case AlzFormat.LZ10: case AlzFormat.LZ11: case AlzFormat.Yaz0: case AlzFormat.Yay0: case AlzFormat.MIO0:
                case AlzFormat.PrsBE: case AlzFormat.PrsLE: case AlzFormat.LZ4Block: case AlzFormat.LZO: case AlzFormat.SnappyRaw:
                case AlzFormat.LZ40: case AlzFormat.CLZ0: case AlzFormat.BLZ: case AlzFormat.LZHudson: break;
                default: return false;
            }
            if (format == AlzFormat.LZSS && lz.HasValue && lz.Value.MaxDistance > 0xFFFF) return false;      // (16-bit links)
            return (uint)sourceLength >= AmdContext.BigStreamCompressThreshold && sourceLength <= 0x20000000
                   && settings.Quality <= AmdContext.BigStreamCompressMaxQuality(format);
        }

        /// <summary>The rule of alz_encode_batch (include/auroralz.h, alz_settings.max_window_bits): 0, or within the format's window
        /// (bits AND 1 &lt;&lt; bits &lt;= the largest distance of the format), or FastLZ up to 20 bits (the device finder keeps 21).</summary>
        internal static bool MaxWindowBitsOnGpu(AlzFormat format, int bits, LzProperties? lz = null)
        {
            if (bits == 0) return true;
            if (format == AlzFormat.FastLZ) return bits <= 20;
            (int wb, int maxDistance) = format switch
            {
                AlzFormat.LZSS => lz.HasValue ? (lz.Value.WindowsBits, lz.Value.MaxDistance) : (12, 0x1000),
                AlzFormat.PrsBE or AlzFormat.PrsLE => (13, 0x1FFF),
                AlzFormat.LZ4Block => (16, 0xFFFF),
                AlzFormat.LZO => (16, 0xBFFF),
                AlzFormat.SnappyRaw => (15, 0x8000),
                _ => (12, 0x1000),                                              // LZ10 / LZ11 / Yaz0 / Yay0 / MIO0
            };
            return bits <= wb && (1 << bits) <= maxDistance;
        }

        internal static AlzLzProperties ToNative(LzProperties lz) => new AlzLzProperties
        {
            WindowBits = lz.WindowsBits, LengthBits = lz.LengthBits, MinLength = (byte)lz.MinLength,
            WindowsStart = (uint)lz.WindowsStart, MaxDistance = (uint)lz.MaxDistance
        };
    }
}

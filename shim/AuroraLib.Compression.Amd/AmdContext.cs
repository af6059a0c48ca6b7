// AmdContext.cs -- one native context per process and device, behind a lock (alz_ctx is single-threaded: one HIP stream).
using System;
using System.Runtime.InteropServices;

namespace AuroraLib.Compression.Amd
{
    /// <summary>
    /// Owner of the native context.  When no HIP device (or no native library) is present, <see cref="Available"/> is false and
    /// every format class of this assembly runs the managed body of the class it mirrors: the native library has no CPU path.
    /// </summary>
    public static class AmdContext
    {
        private static readonly object Gate = new object();
        private static IntPtr _ctx;
        private static bool _probed, _available;

        /// <summary>Streams whose decompressed size is below this run on the managed body: ONE stream is one wavefront --
        /// a lone 256 KiB stream takes ~1.3 ms on the GPU (0.2 GiB/s) against 0.4-0.9 GiB/s of the managed loop
        /// (Benchmarks.md); the GPU pays off for batches (<see cref="BatchDecoder"/>) and for whole files of the framed /
        /// chunked formats, which the library splits into a batch itself.</summary>
        public static uint SingleStreamThreshold { get; set; } = uint.MaxValue;

        /// <summary>LZSS, LZ10, LZ11, Yaz0, Yay0 and MIO0 are the exception (round 4): ONE stream of these formats is a job for the whole
        /// GPU (csrc/alz_big.hip: the group starts of the interleaved formats by list ranking over the input bytes, the cursors of the
        /// three-section formats by prefix sums, the copies by pointer jumping over the output bytes) -- the reference's own benchmark
        /// input, one 1 000 KiB stream of Test.bmp, decodes in 0.27-0.44 ms through <c>alz_decode</c> on host buffers (2.2-3.6 GiB/s)
        /// against 0.32-0.91 GiB/s of the managed loops (Benchmarks.md:30-84).  A single body of these formats with at least this many
        /// decompressed bytes therefore goes to the GPU by default; the native library takes the whole-GPU path from 24 KiB on (against its own wavefront kernels; against the managed loops the call pays from about here).</summary>
        public static uint BigStreamThreshold { get; set; } = 96u << 10;

        /// <summary>The same switch for <c>Compress</c> / <c>CompressHeaderless</c> of ONE buffer: sources shorter than this run on
        /// the managed encoder.  In the batch pipeline the greedy / lazy walk of one stream is a serial job for one wavefront (34 MB:
        /// 1.1 s against 0.13 s on one CPU core, INTEGRATION.md section 1), so the default is "never" for the formats that have no
        /// whole-GPU encode path; the GPU encoder is reached through <see cref="BatchEncoder.CompressMany"/> and through the framed
        /// containers, which split a file into a batch.</summary>
        public static uint SingleStreamCompressThreshold { get; set; } = uint.MaxValue;

        /// <summary>ONE buffer of a format with a whole-GPU encode path (csrc/alz_encode_big.h, round 4: LZSS, LZ10, LZ11, Yaz0, Yay0, MIO0,
        /// PRS, LZ4 blocks, LZO, raw Snappy, LZ40, CLZ0, BLZ, LZHudson) of at least <see cref="BigStreamThreshold"/> bytes is compressed on the
        /// GPU up to this <c>CompressionSettings.Quality</c>: the reference's benchmark input (1 000 KiB of Test.bmp) takes 0.27-0.52 ms
        /// through <c>alz_encode_batch</c> at quality 0 (1.9-3.6 GiB/s against 0.17-0.27 of the managed encoders, Benchmarks.md) and
        /// 0.6-6 ms at quality 15 (0.16-1.7 GiB/s against 0.06-0.13), so the default is "always" (15) for every format of the path.</summary>
        public static int BigStreamCompressMaxQuality(AlzFormat format) => 15;

        /// <summary>... and from this many source bytes on.  The native library takes its whole-GPU encode path from 8 KiB on (against its own
        /// batch pipeline); against the MANAGED encoder -- ~5 ms per MB on one core at quality 0, Benchmarks.md -- a call of ~0.2 ms pays from
        /// about 48 KiB on.</summary>
        public static uint BigStreamCompressThreshold { get; set; } = 64u << 10;

        public static bool Available
        {
            get
            {
                lock (Gate)
                {
                    if (!_probed)
                    {
                        _probed = true;
                        try
                        {
                            _available = Native.alz_abi_version() == 2 && Native.alz_device_count() > 0 && Native.alz_create(0, out _ctx) == 0;
                        }
                        catch (DllNotFoundException) { _available = false; }
                        catch (EntryPointNotFoundException) { _available = false; }
                    }
                    return _available;
                }
            }
        }

        /// <summary>Returns the native context's grow-only device buffers (staging, encoder tables) to the device.</summary>
        public static void ReleaseScratch()
        {
            lock (Gate)
            {
                if (_available) Check(Native.alz_ctx_release_scratch(_ctx));
            }
        }

        internal static IntPtr Handle => Available ? _ctx : throw new InvalidOperationException("no HIP device: " + LastError());
        internal static object Lock => Gate;

        internal static string LastError()
        {
            try { return Marshal.PtrToStringAnsi(Native.alz_last_error()) ?? string.Empty; }
            catch (DllNotFoundException) { return "libauroralz not found"; }
        }

        internal static void Check(int rc)
        {
            if (rc != 0)
                throw new InvalidOperationException("auroralz error " + rc + ": " + LastError());
        }
    }
}

// BatchDecoder.cs -- the entry point the GPU is for: many independent streams in ONE call (one wavefront per stream; a
// 10 000-stream batch fills the device, a single stream uses 1/6000 of it).  The reference issues such batches in
// ScanDecompressCommand (CLI/Commands/ScanDecompressCommand.cs:12-104) and BruteForceCommand (:24-133).
using AuroraLib.Compression.Exceptions;
using System;
using System.Collections.Generic;

namespace AuroraLib.Compression.Amd
{
    /// <summary>One headerless body of a batch: what a format class passes to its static DecompressHeaderless.</summary>
    public readonly struct BodyJob
    {
        public readonly AlzFormat Format;
        public readonly ReadOnlyMemory<byte> Body;
        public readonly uint DecompressedSize;   // decomLength; for bodies without a size field: the output capacity
        public readonly uint Aux0, Aux1;         // Yay0 / MIO0 section pointers relative to the first flag byte
        public BodyJob(AlzFormat format, ReadOnlyMemory<byte> body, uint decompressedSize, uint aux0 = 0, uint aux1 = 0)
        { Format = format; Body = body; DecompressedSize = decompressedSize; Aux0 = aux0; Aux1 = aux1; }
    }

    /// <summary>Result of one body: the bytes produced and the status (0 = the managed body would have returned normally).</summary>
    public readonly struct BodyResult
    {
        public readonly byte[] Data;
        public readonly AlzStatus Status;
        public readonly uint SourceBytesUsed;
        public BodyResult(byte[] data, AlzStatus status, uint used) { Data = data; Status = status; SourceBytesUsed = used; }
        /// <summary>Throws the exception the managed body would have thrown (INTEGRATION.md section 3).</summary>
        public void ThrowIfFailed(long expected) => AmdBody.ThrowForStatus(Status, expected, Data.Length);
    }

    public static unsafe class BatchDecoder
    {
        /// <summary>
        /// Decodes every job on the GPU in one alz_decode_batch call (mixed formats: one kernel per format present, run
        /// concurrently).  LZSS bodies of the batch share <paramref name="lzss"/> (default LzProperties((byte)12, 4, 2), LZSS.cs:33).
        /// </summary>
        public static BodyResult[] DecompressMany(IReadOnlyList<BodyJob> jobs, LzProperties lzss = null)
        {
            int n = jobs.Count;
            var results = new BodyResult[n];
            if (n == 0) return results;
            var streams = new AlzStream[n];
            ulong so = 0, dof = 0;
            for (int i = 0; i < n; i++)
            {
                streams[i] = new AlzStream { SrcOff = so, DstOff = dof, SrcLen = (uint)jobs[i].Body.Length, DstCap = jobs[i].DecompressedSize,
                                             DecomLen = jobs[i].DecompressedSize, Aux0 = jobs[i].Aux0, Aux1 = jobs[i].Aux1, Format = (uint)jobs[i].Format };
                so += ((ulong)jobs[i].Body.Length + 15) & ~15ul;            // inputs packed 16-byte aligned
                dof += ((ulong)jobs[i].DecompressedSize + 255) & ~255ul;     // outputs 256-byte aligned
            }
            if (so > int.MaxValue - 64 || dof > int.MaxValue - 64) throw new NotSupportedException("split the batch: more than 2 GiB in one managed array");
            byte[] src = new byte[so + 64], dst = new byte[dof + 64];
            for (int i = 0; i < n; i++) jobs[i].Body.Span.CopyTo(new Span<byte>(src, (int)streams[i].SrcOff, jobs[i].Body.Length));
            var res = new AlzResult[n];
            AlzLzProperties lz = lzss != null ? AmdBody.ToNative(lzss) : default;
            lock (AmdContext.Lock)
                fixed (byte* ps = src, pd = dst)
                fixed (AlzStream* pst = streams)
                fixed (AlzResult* pr = res)
                    AmdContext.Check(Native.alz_decode_batch(AmdContext.Handle, lzss != null ? &lz : null, (uint)n, ps, (UIntPtr)so, pst, pd, (UIntPtr)dof, pr));
            for (int i = 0; i < n; i++)
            {
                byte[] o = new byte[res[i].DstLen];
                Buffer.BlockCopy(dst, (int)streams[i].DstOff, o, 0, o.Length);
                results[i] = new BodyResult(o, (AlzStatus)res[i].Status, res[i].SrcUsed);
            }
            return results;
        }
    }
}

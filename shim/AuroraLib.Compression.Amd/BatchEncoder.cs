// BatchEncoder.cs -- the compress side of BatchDecoder: many independent buffers through ONE alz_encode_batch call.  One buffer
// is a serial job for one wavefront (the greedy / lazy walk of LzChainMatchFinder.cs:157-212 decides token by token), thousands of
// buffers fill the device: 10 000 x 256 KiB of LZSS take 46 ms at quality 0.  The output of every buffer is bit-identical to what
// the managed CompressHeaderless of its format writes with the same CompressionSettings.
using System;
using System.Collections.Generic;

namespace AuroraLib.Compression.Amd
{
    /// <summary>One raw buffer of a batch and the headerless body it is to become.</summary>
    public readonly struct RawJob
    {
        public readonly AlzFormat Format;
        public readonly ReadOnlyMemory<byte> Data;
        public RawJob(AlzFormat format, ReadOnlyMemory<byte> data) { Format = format; Data = data; }
    }

    /// <summary>The compressed body of one buffer; Yay0 / MIO0: where the token and literal sections start (flags | tokens | literals).</summary>
    public readonly struct EncodedBody
    {
        public readonly byte[] Data;
        public readonly uint Aux0, Aux1;
        public EncodedBody(byte[] data, uint aux0, uint aux1) { Data = data; Aux0 = aux0; Aux1 = aux1; }
    }

    public static unsafe class BatchEncoder
    {
        /// <summary>
        /// Compresses every job on the GPU in one alz_encode_batch call.  <paramref name="settings"/> are the reference's
        /// CompressionSettings (one set for the batch, as one CompressionSettings argument serves one Compress call);
        /// <paramref name="minDistance"/>: 2 = LZ10's GBA VRAM mode (LZ10.cs:25-33), otherwise 0 (the format's own).
        /// LZSS bodies share <paramref name="lzss"/> (default LzProperties((byte)12, 4, 2), LZSS.cs:33).
        /// </summary>
        public static EncodedBody[] CompressMany(IReadOnlyList<RawJob> jobs, CompressionSettings settings = default, int minDistance = 0, LzProperties lzss = null)
        {
            int n = jobs.Count;
            var results = new EncodedBody[n];
            if (n == 0) return results;
            var streams = new AlzStream[n];
            ulong so = 0, dof = 0;
            for (int i = 0; i < n; i++)
            {
                if (!AmdBody.MaxWindowBitsOnGpu(jobs[i].Format, settings.MaxWindowBits))
                    throw new NotSupportedException("MaxWindowBits beyond the format's own window is honoured by the managed encoder only (LzChainMatchFinder.cs:69-73; FastLZ: up to 20 on the GPU)");
                uint len = (uint)jobs[i].Data.Length, cap = len + len / 4 + 64;      // worst case of every body on the path
                streams[i] = new AlzStream { SrcOff = so, DstOff = dof, SrcLen = len, DstCap = cap, Format = (uint)jobs[i].Format };
                so += ((ulong)len + 15) & ~15ul;
                dof += ((ulong)cap + 255) & ~255ul;
            }
            if (so > int.MaxValue - 64 || dof > int.MaxValue - 64) throw new NotSupportedException("split the batch: more than 2 GiB in one managed array");
            byte[] src = new byte[so + 64], dst = new byte[dof + 64];
            for (int i = 0; i < n; i++) jobs[i].Data.Span.CopyTo(new Span<byte>(src, (int)streams[i].SrcOff, jobs[i].Data.Length));
            var res = new AlzResult[n];
            var aux = new AlzEncodeAux[n];
            AlzSettings s = new AlzSettings { Quality = settings.Quality, MaxWindowBits = settings.MaxWindowBits, Strategy = (int)settings.Strategy, MinDistance = minDistance };
            AlzLzProperties lz = lzss != null ? AmdBody.ToNative(lzss) : default;
            lock (AmdContext.Lock)
                fixed (byte* ps = src, pd = dst)
                fixed (AlzStream* pst = streams)
                fixed (AlzResult* pr = res)
                fixed (AlzEncodeAux* pa = aux)
                    AmdContext.Check(Native.alz_encode_batch(AmdContext.Handle, lzss != null ? &lz : null, &s, (uint)n, ps, (UIntPtr)so, pst, pd, (UIntPtr)dof, pr, pa));
            for (int i = 0; i < n; i++)
            {
                if (res[i].Status != (int)AlzStatus.Ok) throw new InvalidOperationException("native encoder status " + res[i].Status + " for job " + i);
                byte[] o = new byte[res[i].DstLen];
                Buffer.BlockCopy(dst, (int)streams[i].DstOff, o, 0, o.Length);
                results[i] = new EncodedBody(o, aux[i].Aux0, aux[i].Aux1);
            }
            return results;
        }
    }
}

// baseline/Program.cs -- the managed CPU baseline of BASELINE.md section 3.1: the reference library itself, timed on the batch
// bench.py measures, with Parallel.For over streams.  Source only: neither the build image nor the GPU box has a .NET SDK
// (probe: `dotnet --info`); a machine that has one runs
//     python tools/export_batch.py --format yaz0 --streams 10000 --stream-kib 256 --out batch.bin      (this repository)
//     dotnet run -c Release --project baseline -- batch.bin [threads]
// and gets "decompressed GiB/s" at one thread and at `threads` (default: all cores) -- the managed counterpart of bench.py's
// cpu_baseline object (which times the C restatement oracle/alz_oracle.c, kind "port").
//
// Protocol = Benchmarks/Benchmarks/TestAllAlgorithms.cs:62-69 of the reference (Instance.Decompress(compressed, output) into
// a pooled stream), batched: every stream of the file is decoded once per pass, passes repeat for >= 10 s.
//
// batch.bin (little endian): "ALZB" u32 version = 1, u32 n, then n records {u32 format (alz_format), u32 decom_len, u32 aux0,
// u32 aux1, u32 src_len}, then the n compressed bodies back to back.
using AuroraLib.Compression.Formats.Common;
using AuroraLib.Compression.Formats.Nintendo;
using AuroraLib.Compression.Formats.Sega;
using AuroraLib.Core.IO;
using System;
using System.Diagnostics;
using System.IO;
using System.Threading.Tasks;

internal static class Program
{
    private struct Item { public uint Format, DecomLen, Aux0, Aux1; public byte[] Body; }

    private static void DecodeOne(in Item it, MemoryPoolStream output)
    {
        using MemoryStream src = new MemoryStream(it.Body, false);
        output.SetLength(0); output.Position = 0;
        switch (it.Format)                                             // the static bodies the GPU path replaces (include/auroralz.h)
        {
            case 0: LZSS.DecompressHeaderless(src, output, it.DecomLen, LZSS.DefaultProperties); break;
            case 1: LZ10.DecompressHeaderless(src, output, it.DecomLen); break;
            case 2: LZ11.DecompressHeaderless(src, output, it.DecomLen); break;
            case 3: Yaz0.DecompressHeaderless(src, output, it.DecomLen); break;
            case 4: Yay0.DecompressHeaderless(src, output, it.DecomLen, (int)it.Aux0, (int)it.Aux1); break;
            case 5: MIO0.DecompressHeaderless(src, output, it.DecomLen, (int)it.Aux0, (int)it.Aux1); break;
            case 6: PRS.DecompressHeaderless(src, output, Endian.Big); break;
            case 7: PRS.DecompressHeaderless(src, output, Endian.Little); break;
            case 8: using (var w = new AuroraLib.Compression.IO.LzWindows(output, 16)) LZ4.DecompressBlockHeaderless(it.Body, w); break;
            case 9: LZO.DecompressHeaderless(src, output); break;
            case 10: Snappy.DecompressHeaderless(src, output); break;
            default: throw new NotSupportedException("format " + it.Format);
        }
        if (it.DecomLen != 0 && output.Length != it.DecomLen) throw new InvalidDataException("size");
    }

    private static double Measure(Item[] items, int threads, double seconds)
    {
        long bytes = 0; int passes = 0;
        var opt = new ParallelOptions { MaxDegreeOfParallelism = threads };
        Stopwatch sw = Stopwatch.StartNew();
        do
        {
            long pass = 0;
            Parallel.For(0, items.Length, opt, () => new MemoryPoolStream(1 << 18), (i, _, o) => { DecodeOne(items[i], o); System.Threading.Interlocked.Add(ref pass, o.Length); return o; }, o => o.Dispose());
            bytes += pass; passes++;
        } while (sw.Elapsed.TotalSeconds < seconds);
        return bytes / sw.Elapsed.TotalSeconds / (1 << 30);
    }

    private static int Main(string[] args)
    {
        if (args.Length < 1) { Console.Error.WriteLine("usage: baseline <batch.bin> [threads]"); return 2; }
        using BinaryReader r = new BinaryReader(File.OpenRead(args[0]));
        if (r.ReadUInt32() != 0x425A4C41u || r.ReadUInt32() != 1) throw new InvalidDataException("not an ALZB version 1 file");
        int n = (int)r.ReadUInt32();
        Item[] items = new Item[n];
        uint[] len = new uint[n];
        for (int i = 0; i < n; i++) { items[i].Format = r.ReadUInt32(); items[i].DecomLen = r.ReadUInt32(); items[i].Aux0 = r.ReadUInt32(); items[i].Aux1 = r.ReadUInt32(); len[i] = r.ReadUInt32(); }
        for (int i = 0; i < n; i++) items[i].Body = r.ReadBytes((int)len[i]);
        int threads = args.Length > 1 ? int.Parse(args[1]) : Environment.ProcessorCount;
        Measure(items, threads, 1.0);                                  // warm-up (JIT, pools)
        double one = Measure(items.AsSpan(0, Math.Min(n, 512)).ToArray(), 1, 10.0);
        double all = Measure(items, threads, 10.0);
        Console.WriteLine("{{\"kind\": \"reference\", \"unit\": \"GiB/s\", \"single_thread\": {0:F3}, \"value\": {1:F3}, \"cores\": {2}, \"streams\": {3}, \"runtime\": \"{4}\"}}",
                          one, all, threads, n, System.Runtime.InteropServices.RuntimeInformation.FrameworkDescription);
        return 0;
    }
}

// Native.cs -- P/Invoke surface of libauroralz.so (include/auroralz.h, ALZ_ABI_VERSION 2).
// Every struct below mirrors a C struct field for field; tests/test_shim_binding.py parses this file and checks each
// [DllImport] (name, parameter count) and each struct (byte size) against the header, because no .NET SDK exists in the
// build image to compile it.  netstandard2.0-safe: UIntPtr instead of nuint, no Stream.ReadExactly, no function pointers.
using System;
using System.Runtime.InteropServices;

namespace AuroraLib.Compression.Amd
{
    /// <summary>alz_lz_properties (16 bytes) -- AuroraLib.Compression.LzProperties (LzProperties.cs:9-97) of the LZSS body.</summary>
    [StructLayout(LayoutKind.Sequential)]
    public struct AlzLzProperties
    {
        public byte WindowBits;
        public byte LengthBits;
        public byte MinLength;
        public byte Reserved0;
        public uint WindowsStart;
        public uint MaxDistance;
        public uint Reserved1;
    }

    /// <summary>alz_stream (40 bytes): one stream of a batch; offsets are relative to the src / dst base of the call.</summary>
    [StructLayout(LayoutKind.Sequential)]
    public struct AlzStream
    {
        public ulong SrcOff;
        public ulong DstOff;
        public uint SrcLen;
        public uint DstCap;
        public uint DecomLen;
        public uint Aux0;
        public uint Aux1;
        public uint Format;
    }

    /// <summary>alz_result (16 bytes).</summary>
    [StructLayout(LayoutKind.Sequential)]
    public struct AlzResult
    {
        public uint DstLen;
        public uint SrcUsed;
        public int Status;
        public uint Reserved;
    }

    /// <summary>alz_settings (16 bytes) -- CompressionSettings (CompressionSettings.cs:11-84).</summary>
    [StructLayout(LayoutKind.Sequential)]
    public struct AlzSettings
    {
        public int Quality;
        public int MaxWindowBits;
        public int Strategy;
        public int MinDistance;
    }

    /// <summary>alz_encode_aux (8 bytes): Yay0 / MIO0 section offsets of an encoded stream.</summary>
    [StructLayout(LayoutKind.Sequential)]
    public struct AlzEncodeAux
    {
        public uint Aux0;
        public uint Aux1;
    }

    /// <summary>alz_format: the headerless bodies (values are ABI constants).</summary>
    public enum AlzFormat : uint
    {
        LZSS = 0, LZ10 = 1, LZ11 = 2, Yaz0 = 3, Yay0 = 4, MIO0 = 5, PrsBE = 6, PrsLE = 7, LZ4Block = 8, LZO = 9, SnappyRaw = 10,
        LZ40 = 11, LZHudson = 12, SMSR00 = 13, FastLZ = 14, CNX2 = 15, BLZ = 16, CLZ0 = 17, CNS = 18, LZ02 = 19, RefPack = 20,
        WFLZ = 21, WFLZ_BE = 22, LZShrek = 23, HIG = 24
    }

    /// <summary>alz_status: the reference's exception classes as per-stream codes (INTEGRATION.md section 3).</summary>
    public enum AlzStatus : int { Ok = 0, InputTruncated = 1, OutputSizeMismatch = 2, OutputCapacity = 3, BadToken = 4 }

    /// <summary>alz_container values used by the framed formats of this assembly.</summary>
    public enum AlzContainer : uint { Snappy = 9, LZ4Frame = 22 }

    internal static unsafe class Native
    {
        private const string Lib = "auroralz";   // libauroralz.so / auroralz.dll on the loader path

        [DllImport(Lib)] internal static extern int alz_abi_version();
        [DllImport(Lib)] internal static extern int alz_device_count();
        [DllImport(Lib)] internal static extern int alz_create(int device, out IntPtr ctx);
        [DllImport(Lib)] internal static extern void alz_destroy(IntPtr ctx);
        [DllImport(Lib)] internal static extern IntPtr alz_last_error();
        [DllImport(Lib)] internal static extern int alz_ctx_set_exact_kernels(IntPtr ctx, int on);
        [DllImport(Lib)] internal static extern int alz_ctx_release_scratch(IntPtr ctx);
        [DllImport(Lib)] internal static extern int alz_ctx_big_stream(IntPtr ctx, uint minBytes, ulong* launchesOut);

        // one stream: backs Decompress(Stream, Stream) of one format class
        [DllImport(Lib)] internal static extern int alz_decode(IntPtr ctx, uint format, AlzLzProperties* props,
            byte* src, uint srcLen, uint decomLen, uint aux0, uint aux1, byte* dst, uint dstCap, AlzResult* result);

        // many streams: the batched form of BruteForceCommand's RawDecoder delegate (CLI/Commands/BruteForceCommand.cs:88-133)
        [DllImport(Lib)] internal static extern int alz_decode_batch(IntPtr ctx, AlzLzProperties* props, uint n,
            byte* srcBase, UIntPtr srcBytes, AlzStream* streams, byte* dstBase, UIntPtr dstBytes, AlzResult* results);

        // the same batch over several contexts (one per GPU), partitioned by the library
        [DllImport(Lib)] internal static extern int alz_decode_batch_multi(IntPtr* ctxs, uint nCtx, AlzLzProperties* props, uint n,
            byte* srcBase, UIntPtr srcBytes, AlzStream* streams, byte* dstBase, UIntPtr dstBytes, AlzResult* results, uint* partOfOut);

        [DllImport(Lib)] internal static extern int alz_partition_batch(uint n, AlzStream* streams, uint nParts, uint* partOf, ulong* partCost);

        // CompressHeaderless of every format class + LzChainMatchFinder (bit-identical to the managed encoder at every quality)
        [DllImport(Lib)] internal static extern int alz_encode_batch(IntPtr ctx, AlzLzProperties* props, AlzSettings* settings, uint n,
            byte* srcBase, UIntPtr srcBytes, AlzStream* streams, byte* dstBase, UIntPtr dstBytes, AlzResult* results, AlzEncodeAux* aux);

        // whole-file helpers of the framed formats (LZ4 frame: descriptor, xxHash32 checksums, linked blocks; Snappy framing: CRC-32C)
        [DllImport(Lib)] internal static extern int alz_container_decompress(IntPtr ctx, uint container, void* opt, byte* src, UIntPtr srcLen,
            byte* dst, UIntPtr dstCap, UIntPtr* dstLen, UIntPtr* srcUsed, int* status);
        [DllImport(Lib)] internal static extern int alz_container_compress(IntPtr ctx, uint container, void* opt, AlzSettings* settings,
            byte* src, UIntPtr srcLen, byte* dst, UIntPtr dstCap, UIntPtr* dstLen);
        [DllImport(Lib)] internal static extern UIntPtr alz_container_compress_bound(uint container, UIntPtr srcLen);
        [DllImport(Lib)] internal static extern int alz_container_decompressed_size(uint container, void* opt, byte* src, UIntPtr srcLen, uint* sizeOut);
    }
}

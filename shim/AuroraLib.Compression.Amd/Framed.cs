// Framed.cs -- LZ4 (frame format) and Snappy (framing format): drop-ins for AuroraLib.Compression.Formats.Common.LZ4 / .Snappy
// (src/AuroraLib.Compression/Formats/Common/LZ4.cs, LZ4.Frame.cs, Snappy.cs).  A frame is a ready-made batch -- every block of
// an LZ4 frame with independent blocks, every 64 KiB chunk of a Snappy file is one GPU stream -- so the whole file goes to
// alz_container_decompress / _compress, which parses the frame on the host (descriptor, xxHash32 / CRC-32C, linked blocks in
// order) and decodes all blocks in ONE launch.  These are the classes where a single Decompress call already fills the GPU.
using AuroraLib.Compression.Interfaces;
using AuroraLib.Core.Format;
using System;
using System.Buffers;
using System.IO;
using Managed = AuroraLib.Compression.Formats.Common;

namespace AuroraLib.Compression.Amd.Common
{
    internal static unsafe class FramedFile
    {
        internal static void Decompress(AlzContainer container, Stream source, Stream destination)
        {
            byte[] src = AmdBody.RentRest(source, out int srcLen);
            try
            {
                uint cap = 0;
                fixed (byte* ps = src)
                    if (Native.alz_container_decompressed_size((uint)container, null, ps, (UIntPtr)(uint)srcLen, &cap) != 0 || cap == 0)
                        cap = (uint)Math.Min(0x7FFF0000L, Math.Max(1 << 20, (long)srcLen * 8));      // no content size in the frame: grow on demand
                for (;;)
                {
                    byte[] dst = ArrayPool<byte>.Shared.Rent((int)cap);
                    try
                    {
                        UIntPtr produced, used; int status, rc;
                        lock (AmdContext.Lock)
                            fixed (byte* ps = src, pd = dst)
                                rc = Native.alz_container_decompress(AmdContext.Handle, (uint)container, null, ps, (UIntPtr)(uint)srcLen, pd, (UIntPtr)cap, &produced, &used, &status);
                        if (rc == -7 && status == (int)AlzStatus.OutputCapacity && cap < 0x7FFF0000u) { cap = cap < 0x3FFF0000u ? cap * 2 : 0x7FFF0000u; continue; }
                        destination.Write(dst, 0, (int)(uint)produced);
                        if (source.CanSeek) source.Position -= srcLen - (int)(uint)used;
                        if (rc == -8) throw new InvalidDataException("checksum mismatch");             // LZ4.Frame.cs:22-28
                        if (rc == -6) throw new AuroraLib.Core.Exceptions.InvalidIdentifierException();
                        if (rc == -7) AmdBody.ThrowForStatus((AlzStatus)status, cap, (long)(uint)produced);
                        AmdContext.Check(rc);
                        return;
                    }
                    finally { ArrayPool<byte>.Shared.Return(dst); }
                }
            }
            finally { ArrayPool<byte>.Shared.Return(src); }
        }

        internal static void Compress(AlzContainer container, ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings)
        {
            uint bound = (uint)Native.alz_container_compress_bound((uint)container, (UIntPtr)(uint)source.Length);
            byte[] dst = ArrayPool<byte>.Shared.Rent((int)bound);
            try
            {
                AlzSettings s = new AlzSettings { Quality = settings.Quality, MaxWindowBits = settings.MaxWindowBits, Strategy = (int)settings.Strategy };
                UIntPtr produced;
                lock (AmdContext.Lock)
                    fixed (byte* ps = source, pd = dst)
                        AmdContext.Check(Native.alz_container_compress(AmdContext.Handle, (uint)container, null, &s, ps, (UIntPtr)(uint)source.Length, pd, (UIntPtr)bound, &produced));
                destination.Write(dst, 0, (int)(uint)produced);
            }
            finally { ArrayPool<byte>.Shared.Return(dst); }
        }
    }

    public sealed class LZ4 : ICompressionAlgorithm
    {
        private static readonly IFormatInfo _info = new FormatInfo<LZ4>("LZ4 Frame Compression (MI355X)", new MediaType(MIMEType.Application, "x-lz4"), ".lz4");

        /// <inheritdoc/>
        public IFormatInfo Info => _info;

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => Managed.LZ4.IsMatchStatic(stream, fileNameAndExtension);   // LZ4.cs:46-48

        /// <inheritdoc/>
        public void Decompress(Stream source, Stream destination)        // LZ4.cs:50-93, LZ4.Frame.cs:107-174
        {
            if (!AmdContext.Available) { new Managed.LZ4().Decompress(source, destination); return; }
            FramedFile.Decompress(AlzContainer.LZ4Frame, source, destination);
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)   // LZ4.cs:113-118, LZ4.Frame.cs:176-215
        {
            if (!AmdContext.Available) { new Managed.LZ4().Compress(source, destination, settings); return; }
            FramedFile.Compress(AlzContainer.LZ4Frame, source, destination, settings);
        }
    }

    public sealed class Snappy : ICompressionAlgorithm
    {
        private static readonly IFormatInfo _info = new FormatInfo<Snappy>("Snappy Frame (MI355X)", new MediaType(MIMEType.Application, "x-snappy-framed"), ".sz");

        /// <inheritdoc/>
        public IFormatInfo Info => _info;

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => Managed.Snappy.IsMatchStatic(stream, fileNameAndExtension);   // Snappy.cs:35-37

        /// <inheritdoc/>
        public void Decompress(Stream source, Stream destination)        // Snappy.cs:39-68
        {
            if (!AmdContext.Available) { new Managed.Snappy().Decompress(source, destination); return; }
            FramedFile.Decompress(AlzContainer.Snappy, source, destination);
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)   // Snappy.cs:71-107
        {
            if (!AmdContext.Available) { new Managed.Snappy().Compress(source, destination, settings); return; }
            FramedFile.Compress(AlzContainer.Snappy, source, destination, settings);
        }
    }
}

// LZSS.cs -- drop-in for AuroraLib.Compression.Formats.Common.LZSS (src/AuroraLib.Compression/Formats/Common/LZSS.cs).
using AuroraLib.Compression.Interfaces;
using AuroraLib.Core.Format;
using AuroraLib.Core.Format.Identifier;
using AuroraLib.Core.IO;
using System;
using System.IO;
using Managed = AuroraLib.Compression.Formats.Common;

namespace AuroraLib.Compression.Amd.Common
{
    public sealed class LZSS : ICompressionAlgorithm, IProvidesDecompressedSize
    {
        private static readonly Identifier32 _identifier = new Identifier32("LZSS".AsSpan());
        private static readonly IFormatInfo _info = new FormatInfo<LZSS>("Lempel-Ziv-Storer-Szymanski (MI355X)", new MediaType(MIMEType.Application, "x-lzss"), string.Empty, _identifier);

        /// <inheritdoc/>
        public IFormatInfo Info => _info;

        private readonly LzProperties LZ;
        public LZSS() : this(Managed.LZSS.DefaultProperties) { }          // LzProperties((byte)12, 4, 2)  LZSS.cs:33
        public LZSS(LzProperties lz) => LZ = lz;

        /// <inheritdoc/>
        public bool IsMatch(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default)
            => stream.Position + 0x10 < stream.Length && stream.Peek(s => s.Match(_identifier));   // LZSS.cs:41-42

        /// <inheritdoc/>
        public uint GetDecompressedSize(Stream source)
            => source.Peek(s => { s.MatchThrow(_identifier); return s.ReadUInt32(Endian.Big); });

        /// <inheritdoc/>
        public void Decompress(Stream source, Stream destination)        // LZSS.cs:53-72
        {
            source.MatchThrow(_identifier);
            uint size = source.ReadUInt32(Endian.Big);
            uint compressedSize = source.ReadUInt32(Endian.Big);
            _ = source.ReadUInt32(Endian.Big);
            _ = compressedSize;                                           // (the reference only traces a mismatch)
            DecompressHeaderless(source, destination, size, LZ);
        }

        /// <summary>LZSS.DecompressHeaderless (LZSS.cs:91-130): the produced size must EQUAL the declared one ('!=', :126).
        /// Windows of 8..16 bits run on the GPU; other geometries stay on the managed body.</summary>
        public static unsafe void DecompressHeaderless(Stream source, Stream destination, uint decomLength, LzProperties lz, byte initialFill = 0x0)
        {
            bool native = initialFill == 0 && lz.WindowsBits >= 8 && lz.WindowsBits <= 16 && lz.LengthBits >= 1 && lz.LengthBits <= 8;
            if (!native || !AmdBody.UseGpuBigStream(decomLength)) { Managed.LZSS.DecompressHeaderless(source, destination, decomLength, lz, initialFill); return; }
            AlzLzProperties p = AmdBody.ToNative(lz);
            AmdBody.Decode(AlzFormat.LZSS, &p, source, destination, decomLength, 0, 0, decomLength + (uint)lz.MaxLength, true);
        }

        /// <inheritdoc/>
        public void Compress(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default)   // LZSS.cs:75-89
        {
            long start = destination.Position;
            destination.Write(_identifier);
            destination.Write(source.Length, Endian.Big);
            destination.Write(0);                                         // compressed length, patched below
            destination.Write(0);
            CompressHeaderless(source, destination, LZ, settings);
            int length = (int)(destination.Position - start - 0x10);
            destination.At(start + 8, x => x.Write(length, Endian.Big));
        }

        /// <summary>LZSS.CompressHeaderless (LZSS.cs:132-160).</summary>
        public static unsafe void CompressHeaderless(ReadOnlySpan<byte> source, Stream destination, LzProperties lz, CompressionSettings settings = default)
        {
            bool native = lz.WindowsBits >= 8 && lz.WindowsBits <= 16 && lz.LengthBits >= 1 && lz.LengthBits <= 8 && lz.MaxDistance == (1 << lz.WindowsBits);
            if (!native || !AmdBody.UseGpuForCompress(AlzFormat.LZSS, source.Length, settings, lz)) { Managed.LZSS.CompressHeaderless(source, destination, lz, settings); return; }
            AlzLzProperties p = AmdBody.ToNative(lz);
            AmdBody.Encode(AlzFormat.LZSS, &p, source, destination, settings, 0);
        }
    }
}

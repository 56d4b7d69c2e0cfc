// What the device-side gzip path (gz_inflate.hip: inflate, CRC-32, FASTA parse on the GPU) and its host driver
// (ingest_gz.cpp) share: the job record of one file, the status codes, and the GF(2) arithmetic of CRC-32 that lets
// every work-item checksum its own span of a text and the spans be put together afterwards.
//
// Reference behaviour this replaces: finch::sketch_files -> needletail::parse_fastx_file auto-detects gzip and inflates
// with flate2's MultiGzDecoder (reference src/finch.rs:69 [SURVEY: recollection of the crate]); a member whose CRC-32 or
// ISIZE does not match is a read error there, hence the checksum here.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define GHIP_GZ_HD __host__ __device__ __forceinline__
#else
#define GHIP_GZ_HD inline
#endif

// status of a file on the device path; anything but GHIP_GZ_OK sends the file through the host path (ingest.cpp), which
// alone decides what is an error and words the message
enum : uint32_t {
    GHIP_GZ_OK = 0,
    GHIP_GZ_EFORMAT = 1,    // not a gzip header this path takes (magic, method, reserved flags, FHCRC, truncated)
    GHIP_GZ_EDATA = 2,      // the deflate stream is damaged (block type 3, over-subscribed code, distance too far back, LEN/NLEN, ...)
    GHIP_GZ_EUNUSUAL = 3,   // legal or possibly legal, but left to the host: an incomplete Huffman code, a code table that outgrew its room
    GHIP_GZ_EOVERFLOW = 4,  // more text than the trailer's ISIZE promised (a wrapped or wrong ISIZE)
    GHIP_GZ_EMULTI = 5,     // bytes behind the first member (further members, or garbage)
    GHIP_GZ_ECRC = 6,       // CRC-32 or ISIZE of the trailer does not match the text
    GHIP_GZ_EFASTA = 7,     // the text is not what the device parser takes: no '>' first, a lone '\r' in front of the first header, too many records
    GHIP_GZ_ENOTRUN = 8,    // (host) the kernels did not get to this file
};

// One gzip file of a batch.  The host fills the first block; gz_inflate_kernel the second; gz_crc_kernel crc_got; the FASTA
// kernels the rest.  in_off / text_off are offsets into the batch's device areas (input images, texts).
struct ghip_gz_job {
    uint64_t in_off;        // the compressed image (a multiple of 16)
    uint64_t text_off;      // where its text goes (a multiple of 64)
    uint64_t gbase;         // base offset of the genome in the resident arrays (a multiple of 64)
    uint32_t in_len;        // bytes of the image
    uint32_t text_cap;      // room for the text = the trailer's ISIZE
    uint32_t stream_cap;    // room for the stream in the resident arrays (the capacity hint the layout was made with)
    uint32_t pad0;
    // ---- gz_inflate_kernel
    uint32_t status;        // GHIP_GZ_*
    uint32_t text_len;
    uint32_t crc_want;      // the trailer's CRC-32
    uint32_t blocks;        // deflate blocks decoded (diagnostics)
    // ---- gz_crc_kernel: XOR of the spans' shifted remainders (ghip_gz::crc_finish turns it into the CRC-32)
    uint32_t crc_acc;
    // ---- FASTA kernels
    uint32_t first_byte;    // index of the first byte that is neither '\n' nor '\r' (text_len: none)
    uint32_t stream_len;    // normalised bytes + one 'N' per record
    uint32_t records;
    uint32_t ambiguous;     // 'N' / 'n' in sequence lines (reference src/genome_stats.rs:26-29)
    uint32_t seq_bytes;     // bytes of sequence lines other than '\n' and '\r': the sum of the record lengths
    uint32_t rec_off;       // this file's record table in the batch's record pool: rec[r] = seq bytes in front of record r
    // ---- diagnostics of the inflate (gz_inflate_kernel): what the time went into
    uint32_t tokens;        // literals + matches decoded
    uint32_t matches;
    uint32_t batches;       // token batches written (<= 64 tokens each)
    uint32_t rounds;        // copy rounds of the batches (a batch with matches takes >= 1)
    uint32_t pad1;
};

namespace ghip_gz {

constexpr uint32_t CRC_POLY = 0xedb88320u;   // CRC-32 (IEEE 802.3), reflected: bit 31 of a word is the coefficient of x^0

// a(x) * b(x) mod P(x)
GHIP_GZ_HD uint32_t gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 31; i >= 0; i--) {
        p ^= b & (0u - ((a >> i) & 1u));
        b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1u)));
    }
    return p;
}

// x^(8 n) mod P: what n further message bytes multiply a remainder by.  x^(2^k) comes from repeated squaring (the order of x
// divides 2^32 - 1, so the exponents wrap at 32 squarings).
GHIP_GZ_HD uint32_t gf_x_pow_bytes(uint64_t n) {
    uint32_t sq = 0x00800000u;   // x^8
    uint32_t p = 0x80000000u;    // x^0
    while (n) {
        if (n & 1u) p = gf_mul(sq, p);
        sq = gf_mul(sq, sq);
        n >>= 1;
    }
    return p;
}

// one message byte into a remainder, bit by bit (the kernels use a table of these for c = 0, byte = 0..255)
GHIP_GZ_HD uint32_t crc_byte(uint32_t c, uint32_t byte) {
    c ^= byte;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (CRC_POLY & (0u - (c & 1u)));
    return c;
}

// The remainder R(M) of a message with register 0 and no final inversion is linear: R(A || B) = R(A) x^(8 |B|) + R(B), and
// leading zero bytes do not change it.  acc = the sum over the spans of R(span) x^(8 bytes behind the span); the CRC-32 of
// gzip (register 0xffffffff first, inverted last) is acc + 0xffffffff x^(8 n) + 0xffffffff.
GHIP_GZ_HD uint32_t crc_finish(uint32_t acc, uint64_t n) { return acc ^ gf_mul(gf_x_pow_bytes(n), 0xffffffffu) ^ 0xffffffffu; }

}  // namespace ghip_gz

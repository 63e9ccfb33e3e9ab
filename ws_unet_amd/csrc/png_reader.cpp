// Host-side batched PNG reader for the evaluate loop (libwsu_io.so; plain C ABI, zlib only, no GPU code).
//
// Why: the predictor runs at > 1.7 k images/s per GPU while the reference's per-image `cv2.imread` + BGR2GRAY
// (src/_defs/imread.py:19-23) -- and PIL in this package -- decode one 512x512 PNG in 2.6-4.7 ms on one core, and Python threads do
// not scale it (the decoders hold the GIL).  wsu_png_read_luma_batch decodes a batch of files on C++ threads straight into one
// (N,H,W) uint8 buffer (the caller passes pinned memory) -- the Y plane that imread4_u8(f)[..., 3] would give:
//   8-bit gray PNG : the stored plane (cv2 replicates gray to BGR and BGR2GRAY maps v,v,v -> v)
//   8-bit RGB PNG  : cv2's fixed-point luma  (R*4899 + G*9617 + B*1868 + 8192) >> 14
// Anything else (palette, alpha, 16 bit, interlaced) reports WSU_PNG_UNSUPPORTED and the Python side reads that file with PIL.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <emmintrin.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

enum { PNG_OK = 0, PNG_IO = -1, PNG_FORMAT = -2, PNG_UNSUPPORTED = -3, PNG_SHAPE = -4 };

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline uint8_t paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

// One scanline: `cur` = the filtered bytes (filter type `ft` already stripped), `prev` = the UNFILTERED previous line or nullptr on the first
// row, result written to `out` (may alias `cur`).  bpp = bytes per pixel (1 gray, 3 RGB).
inline int unfilter_row(int ft, const uint8_t* cur, const uint8_t* prev, uint8_t* out, int stride, int bpp) {
    switch (ft) {
        case 0: if (out != cur) memcpy(out, cur, (size_t)stride); break;
        case 1:
            for (int i = 0; i < bpp; ++i) out[i] = cur[i];
            for (int i = bpp; i < stride; ++i) out[i] = (uint8_t)(cur[i] + out[i - bpp]);
            break;
        case 2:
            if (prev) for (int i = 0; i < stride; ++i) out[i] = (uint8_t)(cur[i] + prev[i]);
            else if (out != cur) memcpy(out, cur, (size_t)stride);
            break;
        case 3:
            for (int i = 0; i < stride; ++i) {
                const int a = i >= bpp ? out[i - bpp] : 0, b = prev ? prev[i] : 0;
                out[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (int i = 0; i < stride; ++i) {
                const int a = i >= bpp ? out[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
                out[i] = (uint8_t)(cur[i] + paeth(a, b, c));
            }
            break;
        default: return PNG_FORMAT;
    }
    return PNG_OK;
}

// The same for 8-bit gray (one byte per pixel, what the data set's files are): Sub as a 16-byte SSE2 prefix sum, Paeth and Average with the
// left neighbour carried in a register and the Paeth predictor written without data-dependent branches.
inline int unfilter_row_gray(int ft, const uint8_t* cur, const uint8_t* prev, uint8_t* out, int n) {
    switch (ft) {
        case 0: memcpy(out, cur, (size_t)n); return PNG_OK;
        case 1: {
            int i = 0;
            __m128i carry = _mm_setzero_si128();
            for (; i + 16 <= n; i += 16) {
                __m128i x = _mm_loadu_si128((const __m128i*)(cur + i));
                x = _mm_add_epi8(x, _mm_slli_si128(x, 1));
                x = _mm_add_epi8(x, _mm_slli_si128(x, 2));
                x = _mm_add_epi8(x, _mm_slli_si128(x, 4));
                x = _mm_add_epi8(x, _mm_slli_si128(x, 8));
                x = _mm_add_epi8(x, carry);
                _mm_storeu_si128((__m128i*)(out + i), x);
                carry = _mm_set1_epi8((char)out[i + 15]);
            }
            uint8_t a = i ? out[i - 1] : 0;
            for (; i < n; ++i) { a = (uint8_t)(cur[i] + a); out[i] = a; }
            return PNG_OK;
        }
        case 2:
            if (!prev) { memcpy(out, cur, (size_t)n); return PNG_OK; }
            for (int i = 0; i < n; ++i) out[i] = (uint8_t)(cur[i] + prev[i]);
            return PNG_OK;
        case 3: {
            int a = 0;
            if (prev) for (int i = 0; i < n; ++i) { a = (uint8_t)(cur[i] + ((a + prev[i]) >> 1)); out[i] = (uint8_t)a; }
            else for (int i = 0; i < n; ++i) { a = (uint8_t)(cur[i] + (a >> 1)); out[i] = (uint8_t)a; }
            return PNG_OK;
        }
        case 4: {
            if (!prev) return unfilter_row_gray(1, cur, nullptr, out, n);      // b = c = 0: the predictor is the left neighbour
            int a = 0, c = 0;
            for (int i = 0; i < n; ++i) {
                const int b = prev[i];
                const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
                int pred = pb <= pc ? b : c;
                pred = (pa <= pb && pa <= pc) ? a : pred;
                a = (uint8_t)(cur[i] + pred);
                out[i] = (uint8_t)a;
                c = b;
            }
            return PNG_OK;
        }
        default: return PNG_FORMAT;
    }
}

// Per-thread scratch, grown once and reused: round 3 allocated three vectors of 170-260 KB per file (file image, concatenated IDAT, the whole
// filtered image) -- each an mmap / page-fault / munmap round trip that serialises the decode threads on the process's memory-map lock, plus a
// second pass over the 262 KB image.  Now: the file image in a reused buffer, the IDAT chunks fed to ONE reused z_stream where they lie, the
// inflated scanlines through a 16-row window (8 KB: stays in L1) and unfiltered STRAIGHT into the destination plane (VERDICT r03 next #7c).
struct Scratch {
    uint8_t* file = nullptr; size_t file_cap = 0;
    uint8_t* win = nullptr; size_t win_cap = 0;            // 16 filtered scanlines
    uint8_t* rgb = nullptr; size_t rgb_cap = 0;            // RGB inputs: two unfiltered lines (current, previous)
    z_stream zs; bool zs_live = false;
    ~Scratch() { free(file); free(win); free(rgb); if (zs_live) inflateEnd(&zs); }
    static bool grow(uint8_t*& p, size_t& cap, size_t need) {
        if (need <= cap) return true;
        uint8_t* q = (uint8_t*)realloc(p, need + need / 4);
        if (!q) return false;
        p = q; cap = need + need / 4;
        return true;
    }
};
constexpr int WIN_ROWS = 16;

// IHDR of a PNG whose first 33 bytes are in `b`: 0 or a negative code
int parse_ihdr(const uint8_t* b, int* w, int* h, int* bpp) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (memcmp(b, sig, 8) != 0 || be32(b + 8) != 13 || memcmp(b + 12, "IHDR", 4) != 0) return PNG_FORMAT;
    *w = (int)be32(b + 16); *h = (int)be32(b + 20);
    const int depth = b[24], color = b[25], interlace = b[28];
    if (depth != 8 || interlace != 0 || (color != 0 && color != 2)) return PNG_UNSUPPORTED;
    if (*w <= 0 || *h <= 0 || *w > 65535 || *h > 65535) return PNG_FORMAT;
    *bpp = color == 0 ? 1 : 3;
    return PNG_OK;
}

// Decode one file into dst (h*w bytes).  expect_h / expect_w > 0: the image must have exactly that shape.
int read_luma(const char* path, uint8_t* dst, int expect_h, int expect_w, int* out_h, int* out_w) {
    thread_local Scratch sc;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return PNG_IO;
    if (!dst) {                                                // header query: 33 bytes, not the file
        uint8_t hd[33];
        const ssize_t got = read(fd, hd, sizeof hd);
        close(fd);
        if (got != (ssize_t)sizeof hd) return got < 0 ? PNG_IO : PNG_FORMAT;
        int w = 0, h = 0, bpp = 0;
        const int rc = parse_ihdr(hd, &w, &h, &bpp);
        if (rc) return rc;
        if (out_h) *out_h = h;
        if (out_w) *out_w = w;
        return ((expect_h > 0 && h != expect_h) || (expect_w > 0 && w != expect_w)) ? PNG_SHAPE : PNG_OK;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8 + 25) { close(fd); return st.st_size < 8 + 25 ? PNG_FORMAT : PNG_IO; }
    const size_t size = (size_t)st.st_size;
    if (!Scratch::grow(sc.file, sc.file_cap, size)) { close(fd); return PNG_IO; }
    size_t have = 0;
    while (have < size) {
        const ssize_t got = read(fd, sc.file + have, size - have);
        if (got <= 0) { close(fd); return PNG_IO; }
        have += (size_t)got;
    }
    close(fd);
    const uint8_t* buf = sc.file;
    int w = 0, h = 0, bpp = 0;
    int rc = parse_ihdr(buf, &w, &h, &bpp);
    if (rc) return rc;
    if (out_h) *out_h = h;
    if (out_w) *out_w = w;
    if ((expect_h > 0 && h != expect_h) || (expect_w > 0 && w != expect_w)) return PNG_SHAPE;
    const int stride = w * bpp, line = stride + 1;
    if (!Scratch::grow(sc.win, sc.win_cap, (size_t)WIN_ROWS * line)) return PNG_IO;
    if (bpp == 3 && !Scratch::grow(sc.rgb, sc.rgb_cap, (size_t)2 * stride)) return PNG_IO;
    if (!sc.zs_live) {
        memset(&sc.zs, 0, sizeof sc.zs);
        if (inflateInit(&sc.zs) != Z_OK) return PNG_IO;
        sc.zs_live = true;
    } else if (inflateReset(&sc.zs) != Z_OK) {
        return PNG_IO;
    }
    z_stream& zs = sc.zs;
    int y = 0;                                                 // next scanline to unfilter
    size_t win_have = 0;                                       // inflated bytes waiting in the window (always whole lines after a drain)
    zs.next_out = sc.win; zs.avail_out = (uInt)((size_t)WIN_ROWS * line);
    auto drain = [&]() -> int {                                // unfilter every whole line in the window into the destination
        const size_t nlines = win_have / (size_t)line;
        if ((size_t)y + nlines > (size_t)h) return PNG_FORMAT;    // more scanlines than IHDR announced
        for (size_t k = 0; k < nlines && y < h; ++k, ++y) {
            const uint8_t* src = sc.win + k * (size_t)line;
            if (bpp == 1) {
                uint8_t* o = dst + (size_t)y * w;
                const int r = unfilter_row_gray(src[0], src + 1, y ? o - w : nullptr, o, w);
                if (r) return r;
            } else {
                uint8_t* cur = sc.rgb + (size_t)(y & 1) * stride;
                const uint8_t* prev = y ? sc.rgb + (size_t)((y - 1) & 1) * stride : nullptr;
                const int r = unfilter_row(src[0], src + 1, prev, cur, stride, 3);
                if (r) return r;
                uint8_t* o = dst + (size_t)y * w;
                for (int x = 0; x < w; ++x)
                    o[x] = (uint8_t)((cur[3 * x] * 4899 + cur[3 * x + 1] * 9617 + cur[3 * x + 2] * 1868 + (1 << 13)) >> 14);
            }
        }
        const size_t used = nlines * (size_t)line;
        if (used < win_have) memmove(sc.win, sc.win + used, win_have - used);
        win_have -= used;
        zs.next_out = sc.win + win_have; zs.avail_out = (uInt)((size_t)WIN_ROWS * line - win_have);
        return PNG_OK;
    };
    size_t pos = 8;
    bool done = false, stream_end = false, any_idat = false;
    while (!done && pos + 12 <= size) {
        const uint32_t len = be32(&buf[pos]);
        const uint8_t* type = &buf[pos + 4];
        const uint8_t* data = &buf[pos + 8];
        if (pos + 12 + (size_t)len > size) return PNG_FORMAT;
        if (!memcmp(type, "IDAT", 4)) {
            any_idat = true;
            zs.next_in = const_cast<Bytef*>(data); zs.avail_in = len;
            while (zs.avail_in > 0 && !stream_end) {
                const uInt before = zs.avail_out;
                const int zr = inflate(&zs, Z_NO_FLUSH);
                win_have += before - zs.avail_out;
                if (zr == Z_STREAM_END) stream_end = true;
                else if (zr != Z_OK && zr != Z_BUF_ERROR) return PNG_FORMAT;
                if (zs.avail_out == 0 || stream_end) { rc = drain(); if (rc) return rc; }
                else if (zr == Z_BUF_ERROR && zs.avail_in > 0) return PNG_FORMAT;      // no progress with input and room left: corrupt
            }
        } else if (!memcmp(type, "IEND", 4)) {
            done = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!any_idat) return PNG_FORMAT;
    rc = drain();
    if (rc) return rc;
    return (y == h && win_have == 0) ? PNG_OK : PNG_FORMAT;
}

}  // namespace

extern "C" {

int wsu_io_version(void) { return 100; }

// Shape of one file (no pixel decode).  Returns 0 or a negative WSU_PNG_* code.
int wsu_png_shape(const char* path, int* h, int* w) { return read_luma(path, nullptr, 0, 0, h, w); }

// Decode n files into dst[n][h][w] on up to nthreads C++ threads.  status[i] = 0 or the negative code of file i (that plane is then
// left untouched).  Returns the number of files that failed.
int wsu_png_read_luma_batch(const char* const* paths, int n, uint8_t* dst, int h, int w, int nthreads, int* status) {
    if (!paths || !dst || !status || n < 0 || h <= 0 || w <= 0) return -1;
    std::atomic<int> next(0), failed(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            status[i] = read_luma(paths[i], dst + (size_t)i * h * w, h, w, nullptr, nullptr);
            if (status[i]) failed.fetch_add(1);
        }
    };
    if (nthreads > n) nthreads = n;
    if (nthreads <= 1) { work(); return failed.load(); }
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    return failed.load();
}

}  // extern "C"

// Host-side batched PNG reader for the evaluate loop (libwsu_io.so; plain C ABI, zlib only, no GPU code).
//
// Why: the predictor runs at > 1.7 k images/s per GPU while the reference's per-image `cv2.imread` + BGR2GRAY
// (src/_defs/imread.py:19-23) -- and PIL in this package -- decode one 512x512 PNG in 2.6-4.7 ms on one core, and Python threads do
// not scale it (the decoders hold the GIL).  wsu_png_read_luma_batch decodes a batch of files on C++ threads straight into one
// (N,H,W) uint8 buffer (the caller passes pinned memory) -- the Y plane that imread4_u8(f)[..., 3] would give:
//   8-bit gray PNG : the stored plane (cv2 replicates gray to BGR and BGR2GRAY maps v,v,v -> v)
//   8-bit RGB PNG  : cv2's fixed-point luma  (R*4899 + G*9617 + B*1868 + 8192) >> 14
// Anything else (palette, alpha, 16 bit, interlaced) reports WSU_PNG_UNSUPPORTED and the Python side reads that file with PIL.
#include <zlib.h>

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

// in-place PNG unfilter of `rows` scanlines of `stride` bytes (each preceded by its filter byte in `raw`)
int unfilter(uint8_t* raw, int rows, int stride, int bpp) {
    const uint8_t* prev = nullptr;
    for (int y = 0; y < rows; ++y) {
        uint8_t* line = raw + (size_t)y * (stride + 1);
        const int ft = line[0];
        uint8_t* cur = line + 1;
        switch (ft) {
            case 0: break;
            case 1: for (int i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
            case 2: if (prev) for (int i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]); break;
            case 3:
                for (int i = 0; i < stride; ++i) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
                    cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
                }
                break;
            case 4:
                for (int i = 0; i < stride; ++i) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
                    cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
                }
                break;
            default: return PNG_FORMAT;
        }
        prev = cur;
    }
    return PNG_OK;
}

// Decode one file into dst (h*w bytes).  expect_h / expect_w > 0: the image must have exactly that shape.
int read_luma(const char* path, uint8_t* dst, int expect_h, int expect_w, int* out_h, int* out_w) {
    FILE* f = fopen(path, "rb");
    if (!f) return PNG_IO;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (size < 8 + 25) { fclose(f); return PNG_FORMAT; }
    std::vector<uint8_t> buf((size_t)size);
    const size_t got = fread(buf.data(), 1, (size_t)size, f);
    fclose(f);
    if (got != (size_t)size) return PNG_IO;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (memcmp(buf.data(), sig, 8) != 0) return PNG_FORMAT;
    size_t pos = 8;
    int w = 0, h = 0, bpp = 0;
    bool have_ihdr = false, done = false;
    std::vector<uint8_t> idat;
    idat.reserve((size_t)size);
    while (!done && pos + 12 <= (size_t)size) {
        const uint32_t len = be32(&buf[pos]);
        const uint8_t* type = &buf[pos + 4];
        const uint8_t* data = &buf[pos + 8];
        if (pos + 12 + (size_t)len > (size_t)size) return PNG_FORMAT;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return PNG_FORMAT;
            w = (int)be32(data); h = (int)be32(data + 4);
            const int depth = data[8], color = data[9], interlace = data[12];
            if (depth != 8 || interlace != 0 || (color != 0 && color != 2)) return PNG_UNSUPPORTED;
            if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return PNG_FORMAT;
            bpp = color == 0 ? 1 : 3;
            have_ihdr = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            done = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || idat.empty()) return PNG_FORMAT;
    if (out_h) *out_h = h;
    if (out_w) *out_w = w;
    if ((expect_h > 0 && h != expect_h) || (expect_w > 0 && w != expect_w)) return PNG_SHAPE;
    if (!dst) return PNG_OK;                                   // header query only
    const int stride = w * bpp;
    std::vector<uint8_t> raw((size_t)h * (stride + 1));
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return PNG_FORMAT;
    const int rc = unfilter(raw.data(), h, stride, bpp);
    if (rc) return rc;
    for (int y = 0; y < h; ++y) {
        const uint8_t* line = raw.data() + (size_t)y * (stride + 1) + 1;
        uint8_t* o = dst + (size_t)y * w;
        if (bpp == 1) memcpy(o, line, (size_t)w);
        else for (int x = 0; x < w; ++x)
            o[x] = (uint8_t)((line[3 * x] * 4899 + line[3 * x + 1] * 9617 + line[3 * x + 2] * 1868 + (1 << 13)) >> 14);
    }
    return PNG_OK;
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

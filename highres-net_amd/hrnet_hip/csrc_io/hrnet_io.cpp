// libhrnet_io.so - host-side input pipeline (include/hrnet_io.h).
// A minimal PNG reader (grayscale, non-interlaced, bit depths 1..16: everything the PROBA-V assets use) on zlib's inflate,
// and the decode -> crop -> float -> pad-to-min_L collate of one batch on a pool of threads.
// Semantics follow src/DataLoader.py:72-148,:195-199 and src/utils.py:85-95 (cited per function); no code of theirs is used.
#include "../../../include/hrnet_io.h"

#include <zlib.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";
void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

struct Png {
    int w = 0, h = 0, depth = 0;
    std::vector<unsigned char> idat;
};

uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// Reads the chunks of `path`; with header_only the IDAT payload is skipped.
int read_chunks(const char* path, Png& png, bool header_only) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_err("cannot open %s", path); return -4; }
    unsigned char sig[8];
    static const unsigned char want[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (fread(sig, 1, 8, f) != 8 || memcmp(sig, want, 8) != 0) { fclose(f); set_err("%s is not a PNG file", path); return -4; }
    bool have_ihdr = false;
    for (;;) {
        unsigned char hd[8];
        if (fread(hd, 1, 8, f) != 8) { fclose(f); set_err("%s: truncated chunk header", path); return -4; }
        const uint32_t len = be32(hd);
        if (memcmp(hd + 4, "IHDR", 4) == 0) {
            unsigned char b[13];
            if (len != 13 || fread(b, 1, 13, f) != 13) { fclose(f); set_err("%s: bad IHDR", path); return -4; }
            png.w = (int)be32(b); png.h = (int)be32(b + 4); png.depth = b[8];
            const int color = b[9], interlace = b[12];
            if (color != 0 || interlace != 0 || b[10] != 0 || b[11] != 0) {
                fclose(f);
                set_err("%s: only non-interlaced grayscale PNGs are supported (colour type %d, interlace %d)", path, color, interlace);
                return -4;
            }
            if (!(png.depth == 1 || png.depth == 2 || png.depth == 4 || png.depth == 8 || png.depth == 16) || png.w <= 0 || png.h <= 0) {
                fclose(f); set_err("%s: unsupported geometry %dx%d depth %d", path, png.w, png.h, png.depth); return -4;
            }
            have_ihdr = true;
            fseek(f, 4, SEEK_CUR);                               // CRC
            if (header_only) { fclose(f); return 0; }
        } else if (memcmp(hd + 4, "IDAT", 4) == 0) {
            const size_t old = png.idat.size();
            png.idat.resize(old + len);
            if (len && fread(png.idat.data() + old, 1, len, f) != len) { fclose(f); set_err("%s: truncated IDAT", path); return -4; }
            fseek(f, 4, SEEK_CUR);
        } else if (memcmp(hd + 4, "IEND", 4) == 0) {
            break;
        } else {
            if (fseek(f, (long)len + 4, SEEK_CUR) != 0) { fclose(f); set_err("%s: truncated chunk", path); return -4; }
        }
    }
    fclose(f);
    if (!have_ihdr) { set_err("%s: no IHDR chunk", path); return -4; }
    return 0;
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Decode rows [r0, r1) x columns [c0, c1) of the image into out (row pitch `pitch` samples).  The whole stream is inflated
// (PNG filters chain rows), the crop is applied while unpacking.
int decode(const char* path, const Png& png, uint16_t* out, int r0, int r1, int c0, int c1, size_t pitch) {
    const int bpp = png.depth == 16 ? 2 : 1;                     // bytes per complete pixel, at least 1 (PNG filter unit)
    const size_t row_bytes = ((size_t)png.w * png.depth + 7) / 8;
    std::vector<unsigned char> raw((row_bytes + 1) * (size_t)png.h);
    uLongf dest_len = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &dest_len, png.idat.data(), (uLong)png.idat.size());
    if (zr != Z_OK || dest_len != raw.size()) { set_err("%s: inflate failed (%d, %lu of %zu bytes)", path, zr, (unsigned long)dest_len, raw.size()); return -4; }
    std::vector<unsigned char> prev(row_bytes, 0);
    for (int y = 0; y < png.h; ++y) {
        unsigned char* line = raw.data() + (size_t)y * (row_bytes + 1);
        const int ft = line[0];
        unsigned char* cur = line + 1;
        switch (ft) {
            case 0: break;
            case 1: for (size_t i = bpp; i < row_bytes; ++i) cur[i] = (unsigned char)(cur[i] + cur[i - bpp]); break;
            case 2: for (size_t i = 0; i < row_bytes; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]); break;
            case 3: for (size_t i = 0; i < row_bytes; ++i) cur[i] = (unsigned char)(cur[i] + (((i >= (size_t)bpp ? cur[i - bpp] : 0) + prev[i]) >> 1)); break;
            case 4: for (size_t i = 0; i < row_bytes; ++i) {
                        const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                        cur[i] = (unsigned char)(cur[i] + paeth(a, b, c));
                    }
                    break;
            default: set_err("%s: bad filter type %d in row %d", path, ft, y); return -4;
        }
        memcpy(prev.data(), cur, row_bytes);
        if (y < r0 || y >= r1) continue;
        uint16_t* o = out + (size_t)(y - r0) * pitch;
        if (png.depth == 16) for (int x = c0; x < c1; ++x) o[x - c0] = (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]);
        else if (png.depth == 8) for (int x = c0; x < c1; ++x) o[x - c0] = cur[x];
        else {
            const int per = 8 / png.depth, mask = (1 << png.depth) - 1;
            for (int x = c0; x < c1; ++x) o[x - c0] = (uint16_t)((cur[x / per] >> ((per - 1 - x % per) * png.depth)) & mask);
        }
    }
    return 0;
}

int read_crop(const char* path, int want_w, int want_h, uint16_t* out, int r0, int r1, int c0, int c1, size_t pitch) {
    Png png;
    int rc = read_chunks(path, png, false);
    if (rc) return rc;
    if (png.w != want_w || png.h != want_h) { set_err("%s is %dx%d, expected %dx%d", path, png.w, png.h, want_w, want_h); return -4; }
    return decode(path, png, out, r0, r1, c0, c1, pitch);
}

}  // namespace

extern "C" {

int hrn_io_version(void) { return 1; }
const char* hrn_io_last_error(void) { return g_err; }

int hrn_io_png_info(const char* path, int* width, int* height, int* bit_depth) {
    if (!path || !width || !height || !bit_depth) { set_err("hrn_io_png_info: null argument"); return -2; }
    Png png;
    const int rc = read_chunks(path, png, true);
    if (rc) return rc;
    *width = png.w; *height = png.h; *bit_depth = png.depth;
    return 0;
}

int hrn_io_png_read_u16(const char* path, uint16_t* out, int width, int height) {
    if (!path || !out || width <= 0 || height <= 0) { set_err("hrn_io_png_read_u16: bad argument"); return -2; }
    return read_crop(path, width, height, out, 0, height, 0, width, (size_t)width);
}

int hrn_io_collate(int n_sets, const char* const* lr_paths, const int* n_views, const char* const* hr_paths,
                   const char* const* sm_paths, int min_L, int lr_size, int patch, const int* px, const int* py,
                   float* lrs, float* alphas, float* hrs, float* maps, int n_threads) {
    if (n_sets <= 0 || !lr_paths || !n_views || !sm_paths || min_L <= 0 || lr_size <= 0 || patch < 0 || !lrs || !alphas || !maps ||
        (patch > 0 && (!px || !py)) || (hr_paths && !hrs)) {
        set_err("hrn_io_collate: bad argument");
        return -2;
    }
    const int S = patch > 0 ? patch : lr_size;
    // work items: (set, view slot) for LR, plus HR and SM per set
    struct Item { int set, kind, slot; const char* path; };      // kind 0 LR, 1 HR, 2 SM
    std::vector<Item> items;
    size_t base = 0;
    for (int s = 0; s < n_sets; ++s) {
        if (n_views[s] < 0) { set_err("hrn_io_collate: negative view count"); return -2; }
        if (patch > 0 && (px[s] < 0 || py[s] < 0 || px[s] + patch > lr_size || py[s] + patch > lr_size)) {
            set_err("hrn_io_collate: patch (%d,%d)+%d outside a %d image", px[s], py[s], patch, lr_size);
            return -2;
        }
        const int used = n_views[s] < min_L ? n_views[s] : min_L;
        for (int v = 0; v < min_L; ++v) {
            alphas[(size_t)s * min_L + v] = v < used ? 1.f : 0.f;                    // utils.py:87-95
            if (v < used) items.push_back({s, 0, v, lr_paths[base + v]});
            else memset(lrs + ((size_t)s * min_L + v) * S * S, 0, sizeof(float) * S * S);
        }
        base += (size_t)n_views[s];
        if (hr_paths && hr_paths[s]) items.push_back({s, 1, 0, hr_paths[s]});
        items.push_back({s, 2, 0, sm_paths[s]});
    }
    std::atomic<size_t> next(0);
    std::atomic<int> status(0);
    std::string first_error;
    std::atomic<bool> have_error(false);
    auto worker = [&]() {
        std::vector<uint16_t> buf((size_t)9 * S * S);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= items.size() || status.load() != 0) return;
            const Item& it = items[i];
            const int x = patch > 0 ? px[it.set] : 0, y = patch > 0 ? py[it.set] : 0;     // x = row corner, y = column corner
            int rc;
            if (it.kind == 0) {
                rc = read_crop(it.path, lr_size, lr_size, buf.data(), x, x + S, y, y + S, (size_t)S);
                if (!rc) {
                    float* o = lrs + ((size_t)it.set * min_L + it.slot) * S * S;
                    for (size_t k = 0; k < (size_t)S * S; ++k) o[k] = (float)((double)buf[k] / 65535.0);   // img_as_float -> float32
                }
            } else {
                const int S3 = 3 * S;
                rc = read_crop(it.path, 3 * lr_size, 3 * lr_size, buf.data(), 3 * x, 3 * x + S3, 3 * y, 3 * y + S3, (size_t)S3);
                if (!rc) {
                    float* o = (it.kind == 1 ? hrs : maps) + (size_t)it.set * S3 * S3;
                    if (it.kind == 1) for (size_t k = 0; k < (size_t)S3 * S3; ++k) o[k] = (float)((double)buf[k] / 65535.0);
                    else for (size_t k = 0; k < (size_t)S3 * S3; ++k) o[k] = buf[k] ? 1.f : 0.f;             // dtype=bool -> float32
                }
            }
            if (rc) {
                bool expected = false;
                if (have_error.compare_exchange_strong(expected, true)) first_error = g_err;
                status.store(rc);
                return;
            }
        }
    };
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if ((size_t)nt > items.size()) nt = (int)items.size();
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    if (status.load() != 0) { set_err("%s", first_error.c_str()); return status.load(); }
    return 0;
}

}  // extern "C"

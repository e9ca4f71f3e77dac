// rf_api_files.hip -- corpus files (save / load / validate) and streamed scans of corpora larger than HBM (split out of rf_api.hip in round 4; rf_host.hpp has the shared declarations).
// Product code: never includes or links anything from oracle/.
#include "rf_host.hpp"

extern "C" {

// ---------------------------------------------------------------------------------------------------
// corpus files, and corpora larger than HBM
// ---------------------------------------------------------------------------------------------------
// The packed form is position-independent (tile descriptors hold offsets, tiles ascend by length), so a corpus can
// be written once and mapped back without re-packing, and ANY tile range [t0, t1) is itself a valid corpus: its
// payload is one contiguous byte range.  rf_stream_many_* uses that to scan a file segment by segment through two
// device buffers, the upload of segment k+1 overlapping the scan of segment k.
namespace {
struct FileHeader {  // little endian, 512 bytes
    char magic[8];   // "RFCORPUS"
    uint32_t version, flags;  // flags: 1 = uniform (no descriptors / orig), 2 = u32 elements (alphabet section)
    uint64_t n;
    uint32_t n_tiles, max_len, uniform_len, n_lengths;
    uint64_t payload_bytes, data_bytes;
    uint64_t off_lengths, off_tiles, off_orig, off_alphabet, off_data;
    uint32_t n_alphabet, n_overflow;
    uint8_t sigma[256];
    uint64_t off_raw;  // flags & 4: the u32 symbol stream parallel to the payload (data_bytes entries)
    uint64_t off_mixed;  // n_mixed MixedDesc, then 64 * n_mixed lengths, then 64 * n_mixed original indices
    uint32_t n_exact, n_mixed;  // tiles [0, n_exact) exact, the rest one-length views of the n_mixed mixed tiles
    uint8_t reserved[512 - 8 - 8 - 8 - 16 - 16 - 40 - 8 - 256 - 8 - 16];
};
static_assert(sizeof(FileHeader) == 512, "header layout");
constexpr uint32_t kFileVersion = 2, kFlagUniform = 1, kFlagWide = 2, kFlagRaw = 4, kFlagRaw16 = 8;

struct FileCloser {
    FILE* f;
    ~FileCloser()
    {
        if (f) std::fclose(f);
    }
};
bool write_all(FILE* f, const void* p, size_t n) { return n == 0 || std::fwrite(p, 1, n, f) == n; }
bool read_at(FILE* f, uint64_t off, void* p, size_t n)
{
    if (n == 0) return true;
    return fseeko(f, (off_t)off, SEEK_SET) == 0 && std::fread(p, 1, n, f) == n;
}
// A segment of the payload, read by up to 8 threads with pread: one thread copies out of the page cache at ~10 GB/s,
// which is what bounded the streamed path.
bool read_parallel(int fd, uint64_t off, uint8_t* dst, size_t n)
{
    // (RF_STREAM_THREADS: reader threads of the streamed scans and of rf_corpus_load; default 16 -- the page-cache -> pinned-buffer copy
    // runs at ~5 GB/s per thread, and it is this copy, not the link, that bounds a streamed scan: profiles/stream_r04.txt)
    static const size_t max_threads = [] { const char* e = getenv("RF_STREAM_THREADS"); const int v = e ? atoi(e) : 16; return (size_t)(v > 0 ? v : 1); }();
    const size_t nthreads = n < (32u << 20) ? 1 : std::min<size_t>(max_threads, std::max<size_t>(1, std::thread::hardware_concurrency()));
    std::atomic<bool> ok{true};
    auto worker = [&](size_t t) {
        size_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
        while (lo < hi) {
            const ssize_t r = pread(fd, dst + lo, hi - lo, (off_t)(off + lo));
            if (r <= 0) {
                ok = false;
                return;
            }
            lo += (size_t)r;
        }
    };
    if (nthreads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nthreads; ++t) pool.emplace_back(worker, t);
        for (auto& th : pool) th.join();
    }
    return ok;
}
rf_status read_header(FILE* f, FileHeader* h)
{
    if (!read_at(f, 0, h, sizeof(*h)) || std::memcmp(h->magic, "RFCORPUS", 8) != 0 || h->version != kFileVersion) {
        set_error("not a corpus file of this version");
        return RF_ERR_INVALID_ARG;
    }
    // every section must lie inside the file (sizes in 128-bit-safe steps: the counts are attacker-sized)
    uint64_t fsize = 0;
    if (fseeko(f, 0, SEEK_END) == 0) fsize = (uint64_t)ftello(f);
    auto inside = [&](uint64_t off, uint64_t count, uint64_t elem) { return off <= fsize && count <= (fsize - off) / std::max<uint64_t>(elem, 1); };
    const bool uniform = (h->flags & kFlagUniform) != 0;
    const uint64_t raw_elem = (h->flags & kFlagRaw) ? ((h->flags & kFlagRaw16) ? 2 : 4) : 0;
    const bool ok = h->n < 0xFFFFFFFFull && inside(h->off_lengths, (uint64_t)h->n_lengths * 2, 4) &&
                    (uniform || (inside(h->off_tiles, h->n_tiles, sizeof(TileDesc)) && inside(h->off_orig, (uint64_t)h->n_tiles * kWave, 4))) &&
                    inside(h->off_alphabet, (uint64_t)h->n_alphabet * 2 + h->n_overflow, 4) && inside(h->off_data, h->data_bytes, 1) &&
                    (!raw_elem || inside(h->off_raw, h->data_bytes, raw_elem)) && h->n_alphabet <= 256 && h->off_data >= sizeof(FileHeader) &&
                    h->n_exact <= h->n_tiles && (h->n_mixed == 0 || (!uniform && inside(h->off_mixed, (uint64_t)h->n_mixed * (sizeof(MixedDesc) + 2 * kWave * 4), 1)));
    if (!ok) {
        set_error("corpus file is inconsistent: a section lies outside the file (truncated?)");
        return RF_ERR_INVALID_ARG;
    }
    return RF_OK;
}
}  // namespace

rf_status rf_corpus_save(const rf_corpus* c, const char* path)
try {
    if (!c || !path || c->borrowed) {
        set_error("rf_corpus_save: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    DeviceGuard guard(c->device);
    if (!guard.ok) return RF_ERR_NO_DEVICE;
    FileCloser fc{std::fopen(path, "wb")};
    if (!fc.f) {
        set_error(std::string("rf_corpus_save: cannot open ") + path);
        return RF_ERR_INVALID_ARG;
    }
    FileHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, "RFCORPUS", 8);
    h.version = kFileVersion;
    h.flags = (c->uniform ? kFlagUniform : 0) | (c->wide ? kFlagWide : 0) | (c->d_raw ? kFlagRaw : 0) | (c->d_raw && c->raw_elem == 2 ? kFlagRaw16 : 0);
    h.n = c->n;
    h.n_tiles = c->n_tiles;
    h.max_len = c->max_len;
    h.uniform_len = c->uniform_len;
    h.n_lengths = (uint32_t)c->lengths.size();
    h.payload_bytes = c->payload_bytes;
    h.data_bytes = c->data_bytes;
    h.n_alphabet = (uint32_t)c->alphabet.size();
    h.n_overflow = (uint32_t)c->overflow.size();
    std::memcpy(h.sigma, c->sigma, 256);
    const size_t n_slots = c->uniform ? 0 : (size_t)c->n_tiles * kWave;
    uint64_t off = sizeof(h);
    h.off_lengths = off, off += (uint64_t)h.n_lengths * 8;
    h.off_tiles = off, off += c->uniform ? 0 : (uint64_t)c->n_tiles * sizeof(TileDesc);
    h.off_orig = off, off += (uint64_t)n_slots * 4;
    h.off_alphabet = off, off += (uint64_t)h.n_alphabet * 8 + (uint64_t)h.n_overflow * 4;
    h.n_exact = c->n_exact;
    h.n_mixed = c->d_mixed ? c->n_mixed : 0;
    h.off_mixed = off, off += (uint64_t)h.n_mixed * (sizeof(MixedDesc) + 2 * kWave * 4);
    h.off_data = (off + 4095) / 4096 * 4096;  // page-aligned payload
    h.off_raw = (h.off_data + c->data_bytes + 4095) / 4096 * 4096;
    bool ok = write_all(fc.f, &h, sizeof(h));
    ok = ok && write_all(fc.f, c->lengths.data(), c->lengths.size() * 4) && write_all(fc.f, c->length_first_tile.data(), c->length_first_tile.size() * 4);
    if (!c->uniform) {
        std::vector<TileDesc> tiles(c->n_tiles);
        std::vector<uint32_t> orig(n_slots);
        RF_HIP(hipMemcpy(tiles.data(), c->d_tiles, tiles.size() * sizeof(TileDesc), hipMemcpyDeviceToHost));
        RF_HIP(hipMemcpy(orig.data(), c->d_orig, orig.size() * 4, hipMemcpyDeviceToHost));
        ok = ok && write_all(fc.f, tiles.data(), tiles.size() * sizeof(TileDesc)) && write_all(fc.f, orig.data(), orig.size() * 4);
    }
    {
        std::vector<uint32_t> a;
        for (const auto& kv : c->alphabet) a.push_back(kv.first), a.push_back(kv.second);
        for (uint32_t sym : c->overflow) a.push_back(sym);
        ok = ok && write_all(fc.f, a.data(), a.size() * 4);
    }
    if (h.n_mixed) {
        std::vector<uint32_t> lens((size_t)h.n_mixed * kWave), origs((size_t)h.n_mixed * kWave);
        RF_HIP(hipMemcpy(lens.data(), c->d_mixed_len, lens.size() * 4, hipMemcpyDeviceToHost));
        RF_HIP(hipMemcpy(origs.data(), c->d_mixed_orig, origs.size() * 4, hipMemcpyDeviceToHost));
        ok = ok && write_all(fc.f, c->mixed.data(), c->mixed.size() * sizeof(MixedDesc)) && write_all(fc.f, lens.data(), lens.size() * 4) &&
             write_all(fc.f, origs.data(), origs.size() * 4);
    }
    std::vector<uint8_t> buf(std::min<uint64_t>(std::max<uint64_t>(c->data_bytes, 1), 64ull << 20));
    ok = ok && fseeko(fc.f, (off_t)h.off_data, SEEK_SET) == 0;
    for (uint64_t done = 0; ok && done < c->data_bytes; done += buf.size()) {
        const size_t m = (size_t)std::min<uint64_t>(buf.size(), c->data_bytes - done);
        RF_HIP(hipMemcpy(buf.data(), c->d_data + done, m, hipMemcpyDeviceToHost));
        ok = write_all(fc.f, buf.data(), m);
    }
    if (c->d_raw) {
        ok = ok && fseeko(fc.f, (off_t)h.off_raw, SEEK_SET) == 0;
        const uint64_t raw_bytes = c->data_bytes * c->raw_elem;
        for (uint64_t done = 0; ok && done < raw_bytes; done += buf.size()) {
            const size_t m = (size_t)std::min<uint64_t>(buf.size(), raw_bytes - done);
            RF_HIP(hipMemcpy(buf.data(), reinterpret_cast<const uint8_t*>(c->d_raw) + done, m, hipMemcpyDeviceToHost));
            ok = write_all(fc.f, buf.data(), m);
        }
    }
    if (!ok || std::fflush(fc.f) != 0) {
        set_error(std::string("rf_corpus_save: write failed: ") + path);
        return RF_ERR_INVALID_ARG;
    }
    return RF_OK;
}
RF_ABI_CATCH

// host-side metadata shared by rf_corpus_load and the stream driver
struct MixedArrays {
    std::vector<uint32_t> len, orig;  // 64 per mixed tile (the descriptors go to rf_corpus::mixed)
};
static rf_status load_meta(FILE* f, const FileHeader& h, rf_corpus* c, std::vector<TileDesc>* tiles, std::vector<uint32_t>* orig, MixedArrays* mx = nullptr)
{
    c->n = h.n;
    c->payload_bytes = h.payload_bytes;
    c->data_bytes = h.data_bytes;
    c->n_tiles = h.n_tiles;
    c->n_exact = h.n_exact;
    c->n_mixed = h.n_mixed;
    c->max_len = h.max_len;
    c->uniform = (h.flags & kFlagUniform) != 0;
    c->uniform_len = h.uniform_len;
    c->wide = (h.flags & kFlagWide) != 0;
    std::memcpy(c->sigma, h.sigma, 256);
    c->lengths.resize(h.n_lengths);
    c->length_first_tile.resize(h.n_lengths);
    bool ok = read_at(f, h.off_lengths, c->lengths.data(), (size_t)h.n_lengths * 4) &&
              read_at(f, h.off_lengths + (uint64_t)h.n_lengths * 4, c->length_first_tile.data(), (size_t)h.n_lengths * 4);
    if (!c->uniform) {
        tiles->resize(h.n_tiles);
        orig->resize((size_t)h.n_tiles * kWave);
        ok = ok && read_at(f, h.off_tiles, tiles->data(), tiles->size() * sizeof(TileDesc)) && read_at(f, h.off_orig, orig->data(), orig->size() * 4);
    }
    std::vector<uint32_t> a((size_t)h.n_alphabet * 2 + h.n_overflow);
    ok = ok && read_at(f, h.off_alphabet, a.data(), a.size() * 4);
    MixedArrays local;
    if (!mx) mx = &local;
    if (h.n_mixed) {
        c->mixed.resize(h.n_mixed);
        mx->len.resize((size_t)h.n_mixed * kWave);
        mx->orig.resize((size_t)h.n_mixed * kWave);
        const uint64_t o1 = h.off_mixed + (uint64_t)h.n_mixed * sizeof(MixedDesc), o2 = o1 + mx->len.size() * 4;
        ok = ok && read_at(f, h.off_mixed, c->mixed.data(), c->mixed.size() * sizeof(MixedDesc)) && read_at(f, o1, mx->len.data(), mx->len.size() * 4) &&
             read_at(f, o2, mx->orig.data(), mx->orig.size() * 4);
    }
    if (!ok) {
        set_error("corpus file truncated");
        return RF_ERR_INVALID_ARG;
    }
    for (uint32_t i = 0; i < h.n_alphabet; ++i) c->alphabet.emplace(a[2 * i], (uint8_t)a[2 * i + 1]);
    for (uint32_t i = 0; i < h.n_overflow; ++i) c->overflow.insert(a[(size_t)h.n_alphabet * 2 + i]);
    // Nothing in the file is trusted: every index the kernels will follow is checked against what it indexes, so a
    // truncated, stale or crafted file is refused here instead of becoming an out-of-bounds device access.
    auto bad = [](const char* what) {
        set_error(std::string("corpus file is inconsistent: ") + what);
        return RF_ERR_INVALID_ARG;
    };
    if (c->data_bytes < kTailPad) return bad("payload smaller than its tail padding");
    const uint64_t body = c->data_bytes - kTailPad;
    if (c->n > (uint64_t)c->n_tiles * kWave) return bad("more candidates than tile slots");
    if ((c->n == 0) != (c->n_tiles == 0) && c->n_tiles == 0) return bad("candidates without tiles");
    if (c->lengths.size() != c->length_first_tile.size()) return bad("length table");
    if (c->n_exact > c->n_tiles) return bad("exact tile count");
    for (size_t i = 0; i < c->lengths.size(); ++i) {
        if (i && c->length_first_tile[i] <= c->length_first_tile[i - 1]) return bad("length table not in tile order");
        if (c->length_first_tile[i] >= c->n_tiles || c->lengths[i] > c->max_len) return bad("length table out of range");
    }
    if (!c->lengths.empty() && c->length_first_tile[0] != 0) return bad("length table does not start at tile 0");
    if (c->uniform) {
        if (c->lengths.size() > 1 || c->uniform_len != c->max_len || (c->n_tiles && c->lengths.empty()) || c->n_mixed || c->n_exact != c->n_tiles)
            return bad("uniform flag vs length table");
        if (tile_bytes(c->uniform_len) > 0xFFFFFFFFull || (uint64_t)c->n_tiles * tile_bytes(c->uniform_len) != body) return bad("uniform payload size");
        if (c->n_tiles && c->n <= (uint64_t)(c->n_tiles - 1) * kWave) return bad("empty trailing tile");
    } else {
        if (c->n_tiles && c->lengths.empty()) return bad("tiles without a length table");
        // exact tiles: ascending lengths, payload blocks back to back; then the mixed blocks, back to back as well; every
        // virtual tile (a one-length view of a mixed tile) must lie inside ONE mixed block and be no longer than it
        size_t li = 0;
        uint64_t expect_off = 0, real = 0;
        for (uint32_t t = 0; t < c->n_tiles; ++t) {
            const TileDesc& td = (*tiles)[t];
            while (li + 1 < c->lengths.size() && c->length_first_tile[li + 1] <= t) ++li;
            if (td.len != c->lengths[li]) return bad("tile length vs length table");
            if (td.slot0 != t * (uint32_t)kWave) return bad("tile slot base");
            if (t < c->n_exact) {
                if (t && td.len < (*tiles)[t - 1].len) return bad("exact tiles not ascending");
                if (td.data_off != expect_off) return bad("tile payload offset");
                expect_off += tile_bytes(td.len);
                if (expect_off > body) return bad("tile payload beyond the data section");
            }
        }
        uint32_t vt = c->n_exact;  // virtual tiles follow their mixed tiles in order
        for (uint32_t m = 0; m < c->n_mixed; ++m) {
            const MixedDesc& md = c->mixed[m];
            if (md.data_off != expect_off || md.min_len > md.max_len || md.max_len > c->max_len || md.slot0 != m * (uint32_t)kWave) return bad("mixed tile descriptor");
            if (m && md.min_len < c->mixed[m - 1].max_len) return bad("mixed tiles not ascending");
            expect_off += tile_bytes(md.max_len);
            if (expect_off > body) return bad("mixed payload beyond the data section");
            uint32_t prev_len = 0;
            bool any = false;
            for (; vt < c->n_tiles && (*tiles)[vt].data_off == md.data_off; ++vt) {
                const TileDesc& td = (*tiles)[vt];
                if (td.len < md.min_len || td.len > md.max_len || (any && td.len <= prev_len)) return bad("view of a mixed tile");
                prev_len = td.len;
                any = true;
            }
            if (!any) return bad("mixed tile without views");
            for (uint32_t r = 0; r < (uint32_t)kWave; ++r) {
                const uint32_t o = mx->orig[(size_t)m * kWave + r], l = mx->len[(size_t)m * kWave + r];
                if (o == kPad) continue;
                if (o >= c->n || l < md.min_len || l > md.max_len) return bad("mixed lane");
            }
        }
        if (vt != c->n_tiles) return bad("views without a mixed tile");
        if (expect_off != body) return bad("data section size");
        std::vector<uint8_t> seen;  // every original index exactly once
        if (c->n <= (64u << 20)) seen.assign((size_t)c->n, 0);
        for (uint32_t v : *orig) {
            if (v == kPad) continue;
            if (v >= c->n) return bad("slot map entry beyond the candidate count");
            if (!seen.empty()) {
                if (seen[v]) return bad("slot map maps two slots to one candidate");
                seen[v] = 1;
            }
            ++real;
        }
        if (real != c->n) return bad("slot map does not cover every candidate");
    }
    for (const auto& kv : c->alphabet)
        if (kv.second >= kOverflowId) return bad("alphabet id");
    return RF_OK;
}

rf_status rf_corpus_load(const char* path, int device, rf_corpus** out)
try {
    if (!path || !out) {
        set_error("rf_corpus_load: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    FileCloser fc{std::fopen(path, "rb")};
    if (!fc.f) {
        set_error(std::string("rf_corpus_load: cannot open ") + path);
        return RF_ERR_INVALID_ARG;
    }
    FileHeader h;
    rf_status s = read_header(fc.f, &h);
    if (s != RF_OK) return s;
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("rf_corpus_load: cannot select device");
        return RF_ERR_NO_DEVICE;
    }
    rf_corpus* c = new (std::nothrow) rf_corpus();
    if (!c) return RF_ERR_OOM;
    c->uid = g_corpus_uid.fetch_add(1);
    c->device = device;
    std::vector<TileDesc> tiles;
    std::vector<uint32_t> orig;
    auto fail = [&](rf_status st) {
        rf_corpus_free(c);
        return st;
    };
    MixedArrays mx;
    s = load_meta(fc.f, h, c, &tiles, &orig, &mx);
    if (s != RF_OK) return fail(s);
    RF_HIP_C(hipMalloc(&c->d_data, std::max<uint64_t>(1, c->data_bytes)));
    std::vector<uint8_t> buf(std::min<uint64_t>(std::max<uint64_t>(c->data_bytes, 1), 64ull << 20));
    uint64_t stored_hist[256] = {0};  // of the STORED symbols (sigma applied), one 64-byte line in 16: sym_freq is not in the file
    for (uint64_t done = 0; done < c->data_bytes; done += buf.size()) {
        const size_t m = (size_t)std::min<uint64_t>(buf.size(), c->data_bytes - done);
        if (!read_at(fc.f, h.off_data + done, buf.data(), m)) {
            set_error("corpus file truncated");
            return fail(RF_ERR_INVALID_ARG);
        }
        for (size_t at = 0; at + 64 <= m; at += 1024)
            for (size_t k = 0; k < 64; ++k) stored_hist[buf[at + k]]++;
        RF_HIP_C(hipMemcpy(c->d_data + done, buf.data(), m, hipMemcpyHostToDevice));
    }
    {   // the symbol frequencies the band prefilter's plan reads (plan_band_filter): a loaded corpus must take the same kernel path
        // as the packed one.  Chunk padding counts as the most frequent symbol's id (0) here -- an over-estimate that can
        // only make the plan more cautious.
        uint64_t by_symbol[256];
        for (int ch = 0; ch < 256; ++ch) by_symbol[ch] = stored_hist[c->sigma[ch]];
        if (!c->wide) symbol_frequencies(by_symbol, c->sym_freq);
    }
    c->device_bytes = c->data_bytes;
    if (h.flags & kFlagRaw) {
        c->raw_elem = (h.flags & kFlagRaw16) ? 2 : 4;
        const uint64_t raw_bytes = c->data_bytes * c->raw_elem;
        RF_HIP_C(hipMalloc(&c->d_raw, std::max<uint64_t>(1, raw_bytes)));
        for (uint64_t done = 0; done < raw_bytes; done += buf.size()) {
            const size_t m = (size_t)std::min<uint64_t>(buf.size(), raw_bytes - done);
            if (!read_at(fc.f, h.off_raw + done, buf.data(), m)) {
                set_error("corpus file truncated");
                return fail(RF_ERR_INVALID_ARG);
            }
            RF_HIP_C(hipMemcpy(reinterpret_cast<uint8_t*>(c->d_raw) + done, buf.data(), m, hipMemcpyHostToDevice));
        }
        c->device_bytes += raw_bytes;
    }
    RF_HIP_C(hipMalloc(&c->d_sigma, 256));
    RF_HIP_C(hipMemcpy(c->d_sigma, c->sigma, 256, hipMemcpyHostToDevice));
    if (!c->uniform) {
        RF_HIP_C(hipMalloc(&c->d_tiles, std::max<size_t>(1, tiles.size()) * sizeof(TileDesc)));
        RF_HIP_C(hipMemcpy(c->d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
        RF_HIP_C(hipMalloc(&c->d_orig, std::max<size_t>(1, orig.size()) * 4));
        RF_HIP_C(hipMemcpy(c->d_orig, orig.data(), orig.size() * 4, hipMemcpyHostToDevice));
        c->n_slots = orig.size();
        if (c->n_exact >= 1024) {
            const std::vector<TileDesc> ordered = tiles_by_origin(tiles, c->n_exact, orig.data());
            RF_HIP_C(hipMalloc(&c->d_tiles_by_origin, ordered.size() * sizeof(TileDesc)));
            RF_HIP_C(hipMemcpy(c->d_tiles_by_origin, ordered.data(), ordered.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
        }
        c->device_bytes += tiles.size() * sizeof(TileDesc) + orig.size() * 4;
    }
    if (c->n_mixed) {
        RF_HIP_C(hipMalloc(&c->d_mixed, c->mixed.size() * sizeof(MixedDesc)));
        RF_HIP_C(hipMemcpy(c->d_mixed, c->mixed.data(), c->mixed.size() * sizeof(MixedDesc), hipMemcpyHostToDevice));
        RF_HIP_C(hipMalloc(&c->d_mixed_len, mx.len.size() * 4));
        RF_HIP_C(hipMemcpy(c->d_mixed_len, mx.len.data(), mx.len.size() * 4, hipMemcpyHostToDevice));
        RF_HIP_C(hipMalloc(&c->d_mixed_orig, mx.orig.size() * 4));
        RF_HIP_C(hipMemcpy(c->d_mixed_orig, mx.orig.data(), mx.orig.size() * 4, hipMemcpyHostToDevice));
        c->device_bytes += c->mixed.size() * sizeof(MixedDesc) + 2 * mx.len.size() * 4;
    }
    *out = c;
    return RF_OK;
}
RF_ABI_CATCH

// The streamed scans' buffer sets (pinned host + device payload buffers), kept between calls per process (stream_many below says why);
// rf_release_caches() gives them back.
struct KeptSets {
    std::mutex mu;
    uint8_t *d_data[3] = {nullptr, nullptr, nullptr}, *h_data[3] = {nullptr, nullptr, nullptr};
    uint64_t cap = 0;
    int device = -1;
};
static KeptSets& kept_sets()
{
    static KeptSets k;
    return k;
}

// One pass of `scorer.<op>` over a corpus FILE that need not fit in HBM.  Segments are tile ranges of at most
// `segment_bytes` of payload; two device buffer sets alternate, segment k+1 is read and uploaded (copy stream) while
// segment k is scanned (compute stream).  The result vector (n x 4 or 8 bytes) does live on the device for the pass.
static rf_status stream_many(const rf_comparator* c, const char* path, rf_op op, const rf_args* args, void* out_host, size_t out_capacity,
                             bool f64_out, uint64_t segment_bytes, int device)
{
    const rf_args args_v = args ? sanitized_args(args, false) : rf_args{};  // (a streamed scan's segments have no slot order of their own to offer)
    if (args) args = &args_v;
    if (!c || !path || !args || !out_host) {
        set_error("rf_stream_many: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    const auto t_enter = std::chrono::steady_clock::now();
    FileCloser fc{std::fopen(path, "rb")};
    if (!fc.f) {
        set_error(std::string("rf_stream_many: cannot open ") + path);
        return RF_ERR_INVALID_ARG;
    }
    FileHeader h;
    rf_status s = read_header(fc.f, &h);
    if (s != RF_OK) return s;
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("rf_stream_many: cannot select device");
        return RF_ERR_NO_DEVICE;
    }
    const int fd = fileno(fc.f);  // payload reads go through pread on several threads (read_parallel)
    rf_corpus meta;  // whole-file metadata (host side only)
    meta.uid = g_corpus_uid.fetch_add(1);
    meta.device = device;
    std::vector<TileDesc> tiles;
    std::vector<uint32_t> orig;
    s = load_meta(fc.f, h, &meta, &tiles, &orig);
    if (s != RF_OK) return s;
    if (out_capacity < meta.n) {
        set_error("rf_stream_many: the file holds " + std::to_string(meta.n) + " candidates but `out` has room for " + std::to_string(out_capacity) +
                  " (ask rf_corpus_file_count)");
        return RF_ERR_INVALID_ARG;
    }
    if (meta.n == 0) return RF_OK;
    const uint64_t uniform_tb = tile_bytes(meta.uniform_len);
    auto tile_off = [&](uint32_t t) { return meta.uniform ? (uint64_t)t * uniform_tb : (t < meta.n_tiles ? tiles[t].data_off : meta.data_bytes - kTailPad); };
    // segment boundaries
    if (segment_bytes == 0) segment_bytes = 512ull << 20;  // (64 GB file, same box: 256 MiB segments 39.8 GB/s, 512 MiB 47.5: profiles/stream_r04.txt)
    std::vector<uint32_t> cuts{0};
    // (the one-length views of a mixed tile share one payload block: a cut may only fall where the payload offset changes)
    auto block_start = [&](uint32_t t) { return t >= meta.n_tiles || t == 0 || tile_off(t) != tile_off(t - 1); };
    while (cuts.back() < meta.n_tiles) {
        uint32_t t0 = cuts.back(), t1 = t0 + 1;
        while (t1 < meta.n_tiles && !block_start(t1)) ++t1;  // at least one whole block
        while (t1 < meta.n_tiles) {
            uint32_t t2 = t1 + 1;
            while (t2 < meta.n_tiles && !block_start(t2)) ++t2;
            if (tile_off(t2) - tile_off(t0) > segment_bytes) break;
            t1 = t2;
        }
        cuts.push_back(t1);
    }
    uint64_t max_seg = 0;
    uint32_t max_tiles = 0;
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
        max_seg = std::max(max_seg, tile_off(cuts[k + 1]) - tile_off(cuts[k]));
        max_tiles = std::max(max_tiles, cuts[k + 1] - cuts[k]);
    }

    // A u32 query with a symbol of the file's overflow class is served from a per-segment translated image (make_effective): then the
    // raw symbol stream of each segment travels with its payload (2 or 4 more bytes per stored symbol on the link, for those queries only).
    const uint64_t raw_elem = (h.flags & kFlagRaw) ? ((h.flags & kFlagRaw16) ? 2 : 4) : 0;
    bool need_raw = false;
    if (meta.wide && raw_elem) {
        const rf_comparator* eff = nullptr;
        ComparatorRef hold;
        bool hit = false;
        need_raw = resolve(c, &meta, &eff, &hold, &hit) != RF_OK && hit;
    }

    constexpr int kSlots = 3;  // buffer sets in rotation: one being read into, one on the link, one being scanned
    struct Slot {
        uint8_t *d_data = nullptr, *h_data = nullptr;
        uint8_t *d_raw = nullptr, *h_raw = nullptr;
        TileDesc* d_tiles = nullptr;
        uint32_t* d_orig = nullptr;
        hipEvent_t uploaded = nullptr, scanned = nullptr;
        bool used = false;
    } slot[kSlots];
    hipStream_t s_copy = nullptr, s_comp = nullptr;
    uint8_t *d_sigma = nullptr, *d_identity = nullptr;
    void* d_out = nullptr;
    const size_t elem = f64_out ? sizeof(double) : sizeof(uint32_t);
    rf_status status = RF_OK;
    hipError_t e = hipSuccess;
    auto hip_ok = [&](hipError_t err) {
        if (err != hipSuccess && e == hipSuccess) e = err;
        return err == hipSuccess;
    };
    bool ok = hip_ok(hipStreamCreate(&s_copy)) && hip_ok(hipStreamCreate(&s_comp)) && hip_ok(hipMalloc(&d_sigma, 256)) &&
              hip_ok(hipMemcpy(d_sigma, meta.sigma, 256, hipMemcpyHostToDevice)) && hip_ok(hipMalloc(&d_out, meta.n * elem));
    if (ok && need_raw) {
        uint8_t ident[256];
        for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
        ok = hip_ok(hipMalloc(&d_identity, 256)) && hip_ok(hipMemcpy(d_identity, ident, 256, hipMemcpyHostToDevice));
    }
    // None everywhere first: a cutoff run skips whole tile ranges (plan()), and segment views never pre-fill
    if (ok) ok = hip_ok(hipMemsetAsync(d_out, 0xFF, meta.n * elem, s_comp));
    // The buffer sets (pinned host + device payload buffers) are KEPT between calls, per process: allocating and pinning 3 x 256 MiB
    // costs 50-80 ms, a third of a 6.4 GB streamed scan (profiles/stream_r04.txt).  One streamed scan at a time uses the kept sets (a
    // concurrent one allocates its own); a call that needs larger segments replaces them.  RF_STREAM_KEEP=0: allocate and free per call.
    KeptSets& kept = kept_sets();
    static const bool keep_sets = [] { const char* e = getenv("RF_STREAM_KEEP"); return !e || atoi(e) != 0; }();
    std::unique_lock<std::mutex> kept_lock(kept.mu, std::defer_lock);
    const bool use_kept = keep_sets && kept_lock.try_lock();
    if (use_kept && (kept.cap < max_seg + kTailPad || kept.device != device)) {
        for (int b = 0; b < kSlots; ++b) {
            if (kept.d_data[b]) (void)hipFree(kept.d_data[b]);
            if (kept.h_data[b]) (void)hipHostFree(kept.h_data[b]);
            kept.d_data[b] = kept.h_data[b] = nullptr;
        }
        kept.cap = 0;
        kept.device = device;
        bool got = true;
        for (int b = 0; got && b < kSlots; ++b)
            got = hipMalloc(&kept.d_data[b], max_seg + kTailPad) == hipSuccess && hipHostMalloc((void**)&kept.h_data[b], max_seg + kTailPad, hipHostMallocDefault) == hipSuccess;
        if (got) {
            kept.cap = max_seg + kTailPad;
        } else {
            (void)hipGetLastError();
            for (int b = 0; b < kSlots; ++b) {
                if (kept.d_data[b]) (void)hipFree(kept.d_data[b]);
                if (kept.h_data[b]) (void)hipHostFree(kept.h_data[b]);
                kept.d_data[b] = kept.h_data[b] = nullptr;
            }
        }
    }
    const bool from_kept = use_kept && kept.cap >= max_seg + kTailPad;
    for (int b = 0; ok && b < kSlots; ++b) {
        if (from_kept) {
            slot[b].d_data = kept.d_data[b];
            slot[b].h_data = kept.h_data[b];
        } else {
            ok = hip_ok(hipMalloc(&slot[b].d_data, max_seg + kTailPad)) && hip_ok(hipHostMalloc((void**)&slot[b].h_data, max_seg + kTailPad, hipHostMallocDefault));
        }
        ok = ok && hip_ok(hipEventCreateWithFlags(&slot[b].uploaded, hipEventDisableTiming)) && hip_ok(hipEventCreateWithFlags(&slot[b].scanned, hipEventDisableTiming));
        if (ok && need_raw)
            ok = hip_ok(hipMalloc(&slot[b].d_raw, (max_seg + kTailPad) * raw_elem)) && hip_ok(hipHostMalloc((void**)&slot[b].h_raw, (max_seg + kTailPad) * raw_elem, hipHostMallocDefault));
        if (ok && !meta.uniform)
            ok = hip_ok(hipMalloc(&slot[b].d_tiles, (size_t)max_tiles * sizeof(TileDesc))) && hip_ok(hipMalloc(&slot[b].d_orig, (size_t)max_tiles * kWave * 4));
    }
    // Single-length corpora: a segment's results are a contiguous slice of `out`, so they travel to the host while the later segments
    // are still being read, copied and scanned -- on a thread of their own (a device-to-pageable-host copy blocks its caller) and a
    // stream of their own (the link is full duplex).  4 bytes per 64-byte candidate: left to the end they were 15 % of a 64 GB scan.
    struct ResJob {
        hipEvent_t ready;
        size_t off, bytes;
    };
    std::mutex res_mu;
    std::condition_variable res_cv;
    std::deque<ResJob> res_jobs;
    bool res_done = false;
    std::atomic<int> res_err{(int)hipSuccess};
    hipStream_t s_res = nullptr;
    std::thread res_thread;
    // From here to the join below nothing may unwind past the result thread (a joinable std::thread destroyed during unwinding is
    // std::terminate) or past the buffers, events and streams released after it: an exception (bad_alloc from the per-segment vectors, a
    // std::string of set_error) is parked, the common clean-up runs, and it is rethrown for RF_ABI_CATCH to translate.
    std::exception_ptr pending;
    bool early_results = false;
    const auto t_loop = std::chrono::steady_clock::now();
    double s_wait = 0.0, s_read = 0.0;
    try {
    early_results = ok && meta.uniform && hip_ok(hipStreamCreateWithFlags(&s_res, hipStreamNonBlocking));
    if (early_results)
        res_thread = std::thread([&] {
            (void)hipSetDevice(device);
            while (true) {
                ResJob j;
                {
                    std::unique_lock<std::mutex> lk(res_mu);
                    res_cv.wait(lk, [&] { return res_done || !res_jobs.empty(); });
                    if (res_jobs.empty()) return;
                    j = res_jobs.front();
                    res_jobs.pop_front();
                }
                hipError_t er = hipStreamWaitEvent(s_res, j.ready, 0);
                if (er == hipSuccess) er = hipMemcpyAsync(static_cast<char*>(out_host) + j.off, static_cast<char*>(d_out) + j.off, j.bytes, hipMemcpyDeviceToHost, s_res);
                if (er == hipSuccess) er = hipStreamSynchronize(s_res);
                (void)hipEventDestroy(j.ready);
                if (er != hipSuccess) res_err = (int)er;
            }
        });
    std::vector<TileDesc> seg_tiles;
    static const bool stream_timing = getenv("RF_STREAM_TIMING") != nullptr;  // phase times of a streamed scan on stderr
    using clk = std::chrono::steady_clock;
    if (stream_timing) std::fprintf(stderr, "[rf stream] set-up (streams, device + pinned buffers, None pre-fill) %.1f ms\n", std::chrono::duration<double, std::milli>(t_loop - t_enter).count());
    for (size_t k = 0; ok && status == RF_OK && k + 1 < cuts.size(); ++k) {
        Slot& sl = slot[k % kSlots];
        const uint32_t t0 = cuts[k], t1 = cuts[k + 1];
        const uint64_t base = tile_off(t0), bytes = tile_off(t1) - base;
        const auto t_a = clk::now();
        if (sl.used) ok = hip_ok(hipEventSynchronize(sl.scanned));  // the scan that last read this buffer set is done
        if (!ok) break;
        const auto t_b = clk::now();
        if (!read_parallel(fd, h.off_data + base, sl.h_data, (size_t)bytes)) {
            set_error("corpus file truncated");
            status = RF_ERR_INVALID_ARG;
            break;
        }
        s_wait += std::chrono::duration<double, std::milli>(t_b - t_a).count();
        s_read += std::chrono::duration<double, std::milli>(clk::now() - t_b).count();
        std::memset(sl.h_data + bytes, 0, kTailPad);
        ok = hip_ok(hipMemcpyAsync(sl.d_data, sl.h_data, bytes + kTailPad, hipMemcpyHostToDevice, s_copy));
        if (ok && need_raw) {
            if (!read_parallel(fd, h.off_raw + base * raw_elem, sl.h_raw, (size_t)(bytes * raw_elem))) {
                set_error("corpus file truncated");
                status = RF_ERR_INVALID_ARG;
                break;
            }
            std::memset(sl.h_raw + bytes * raw_elem, 0xFF, kTailPad * raw_elem);  // the raw stream's padding value
            ok = hip_ok(hipMemcpyAsync(sl.d_raw, sl.h_raw, (bytes + kTailPad) * raw_elem, hipMemcpyHostToDevice, s_copy));
        }
        rf_corpus seg;  // a view: owns nothing
        seg.borrowed = true;
        seg.no_prefill = true;
        seg.uid = meta.uid;  // one lowered comparator serves every segment of a u32 corpus
        seg.device = device;
        seg.wide = meta.wide;
        seg.alphabet = meta.alphabet;
        seg.overflow = meta.overflow;
        std::memcpy(seg.sigma, meta.sigma, 256);
        seg.d_sigma = d_sigma;
        seg.d_data = sl.d_data;
        if (need_raw) {
            seg.d_raw = sl.d_raw;
            seg.raw_elem = (uint32_t)raw_elem;
            seg.d_sigma_identity = d_identity;
        }
        seg.n_tiles = t1 - t0;
        seg.n_exact = seg.n_tiles;  // (a segment view has no mixed section of its own: mixed tiles are scanned through their views)
        seg.data_bytes = bytes + kTailPad;
        void* seg_out = d_out;
        if (meta.uniform) {
            seg.uniform = true;
            seg.uniform_len = meta.uniform_len;
            seg.max_len = meta.uniform_len;
            seg.n = (size_t)std::min<uint64_t>((uint64_t)(t1 - t0) * kWave, meta.n - (uint64_t)t0 * kWave);
            seg.lengths = {meta.uniform_len};
            seg.length_first_tile = {0};
            seg_out = static_cast<char*>(d_out) + (size_t)t0 * kWave * elem;  // slot == original index
        } else {
            seg.n = meta.n;  // results are scattered through orig[] into the whole output
            seg_tiles.assign(tiles.begin() + t0, tiles.begin() + t1);
            for (uint32_t t = 0; t < t1 - t0; ++t) {
                seg_tiles[t].data_off -= base;
                seg_tiles[t].slot0 = t * kWave;
                if (seg.lengths.empty() || seg.lengths.back() != seg_tiles[t].len) {
                    seg.lengths.push_back(seg_tiles[t].len);
                    seg.length_first_tile.push_back(t);
                }
                seg.max_len = std::max(seg.max_len, seg_tiles[t].len);
            }
            seg.d_tiles = sl.d_tiles;
            seg.d_orig = sl.d_orig;
            // pageable sources: both copies are staged before the calls return, so seg_tiles may be reused
            ok = ok && hip_ok(hipMemcpyAsync(sl.d_tiles, seg_tiles.data(), seg_tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, s_copy)) &&
                 hip_ok(hipMemcpyAsync(sl.d_orig, orig.data() + (size_t)t0 * kWave, (size_t)(t1 - t0) * kWave * 4, hipMemcpyHostToDevice, s_copy));
        }
        ok = ok && hip_ok(hipEventRecord(sl.uploaded, s_copy)) && hip_ok(hipStreamWaitEvent(s_comp, sl.uploaded, 0));
        if (!ok) break;
        status = run_many(c, &seg, op, args, seg_out, RF_MEM_DEVICE, s_comp, f64_out);
        if (status != RF_OK) break;
        ok = hip_ok(hipEventRecord(sl.scanned, s_comp));
        sl.used = true;
        if (ok && early_results) {
            ResJob j{nullptr, (size_t)t0 * kWave * elem, seg.n * elem};
            ok = hip_ok(hipEventCreateWithFlags(&j.ready, hipEventDisableTiming)) && hip_ok(hipEventRecord(j.ready, s_comp));
            if (ok) {
                std::lock_guard<std::mutex> lk(res_mu);
                res_jobs.push_back(j);
            }
            res_cv.notify_one();
        }
    }
    } catch (...) {
        pending = std::current_exception();
    }
    using clk = std::chrono::steady_clock;
    static const bool stream_timing = getenv("RF_STREAM_TIMING") != nullptr;  // phase times of a streamed scan on stderr
    const auto t_tail = clk::now();
    if (res_thread.joinable()) {
        {
            std::lock_guard<std::mutex> lk(res_mu);
            res_done = true;
        }
        res_cv.notify_one();
        res_thread.join();
        if (res_err.load() != (int)hipSuccess) ok = hip_ok((hipError_t)res_err.load());
    }
    if (!pending && ok && status == RF_OK && !early_results)
        ok = hip_ok(hipMemcpyAsync(out_host, d_out, meta.n * elem, hipMemcpyDeviceToHost, s_comp)) && hip_ok(hipStreamSynchronize(s_comp));
    if (stream_timing)
        std::fprintf(stderr, "[rf stream] %zu segments: loop %.1f ms (reads %.1f, waits for a free buffer set %.1f), drain + results to the host %.1f ms\n", cuts.size() - 1,
                     std::chrono::duration<double, std::milli>(t_tail - t_loop).count(), s_read, s_wait, std::chrono::duration<double, std::milli>(clk::now() - t_tail).count());
    if (s_copy) (void)hipStreamSynchronize(s_copy);
    if (s_comp) (void)hipStreamSynchronize(s_comp);
    for (int b = 0; b < kSlots; ++b) {
        if (slot[b].d_data && !from_kept) (void)hipFree(slot[b].d_data);
        if (slot[b].h_data && !from_kept) (void)hipHostFree(slot[b].h_data);
        if (slot[b].d_raw) (void)hipFree(slot[b].d_raw);
        if (slot[b].h_raw) (void)hipHostFree(slot[b].h_raw);
        if (slot[b].d_tiles) (void)hipFree(slot[b].d_tiles);
        if (slot[b].d_orig) (void)hipFree(slot[b].d_orig);
        if (slot[b].uploaded) (void)hipEventDestroy(slot[b].uploaded);
        if (slot[b].scanned) (void)hipEventDestroy(slot[b].scanned);
    }
    {   // this pass' identity dies with it: drop the comparator lowered for it (a long-lived u32 comparator would
        // otherwise accumulate one cache entry per streamed pass)
        std::lock_guard<std::mutex> lock(c->mu);
        c->lowered.erase(meta.uid);
    }
    if (d_sigma) (void)hipFree(d_sigma);
    if (d_identity) (void)hipFree(d_identity);
    if (d_out) (void)hipFree(d_out);
    if (s_copy) (void)hipStreamDestroy(s_copy);
    if (s_comp) (void)hipStreamDestroy(s_comp);
    if (s_res) (void)hipStreamDestroy(s_res);
    if (pending) std::rethrow_exception(pending);
    if (status != RF_OK) return status;
    if (!ok) {
        set_error(std::string("rf_stream_many: ") + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? RF_ERR_OOM : RF_ERR_HIP;
    }
    return RF_OK;
}

rf_status rf_stream_many_u32(const rf_comparator* c, const char* path, rf_op op, const rf_args* args, uint32_t* out, size_t out_capacity,
                             uint64_t segment_bytes, int device)
try {
    return stream_many(c, path, op, args, out, out_capacity, false, segment_bytes, device);
}
RF_ABI_CATCH
rf_status rf_stream_many_f64(const rf_comparator* c, const char* path, rf_op op, const rf_args* args, double* out, size_t out_capacity,
                             uint64_t segment_bytes, int device)
try {
    return stream_many(c, path, op, args, out, out_capacity, true, segment_bytes, device);
}
RF_ABI_CATCH
rf_status rf_release_caches(void)
try {
    {   // the kept buffer sets of the streamed scans (a scan in flight holds the lock: its sets stay)
        KeptSets& kept = kept_sets();
        std::unique_lock<std::mutex> lk(kept.mu, std::try_to_lock);
        if (lk.owns_lock()) {
            int prev = 0;
            const bool switched = kept.device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != kept.device && hipSetDevice(kept.device) == hipSuccess;
            for (int b = 0; b < 3; ++b) {
                if (kept.d_data[b]) (void)hipFree(kept.d_data[b]);
                if (kept.h_data[b]) (void)hipHostFree(kept.h_data[b]);
                kept.d_data[b] = kept.h_data[b] = nullptr;
            }
            kept.cap = 0;
            kept.device = -1;
            if (switched) (void)hipSetDevice(prev);
        }
    }
    scratch_trim();  // parked blocks of the stream-ordered scratch allocator whose work is done
    return RF_OK;
}
RF_ABI_CATCH
rf_status rf_corpus_file_count(const char* path, size_t* n)
try {
    if (!path || !n) {
        set_error("rf_corpus_file_count: invalid argument");
        return RF_ERR_INVALID_ARG;
    }
    FileCloser fc{std::fopen(path, "rb")};
    if (!fc.f) {
        set_error(std::string("rf_corpus_file_count: cannot open ") + path);
        return RF_ERR_INVALID_ARG;
    }
    FileHeader h;
    const rf_status s = read_header(fc.f, &h);
    if (s == RF_OK) *n = (size_t)h.n;
    return s;
}
RF_ABI_CATCH


}  // extern "C"

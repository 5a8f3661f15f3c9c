// api_ingest.h -- C ABI: CSV ingest on the host (sprk_pack_csv[_mt]) and on the device (sprk_pack_csv_device), sprk_cross_hash.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---- host ingest: CSV text -> packed ids / dense (schema.py's read_samples_csv + pack_ids + pack_dense in one pass) ----
namespace {
const char* const kGenreVocab[19] = {"Film-Noir", "Action", "Adventure", "Horror", "Romance", "War", "Comedy", "Western",
                                     "Documentary", "Sci-Fi", "Drama", "Thriller", "Crime", "Fantasy", "Animation", "IMAX",
                                     "Mystery", "Children", "Musical"};   // DeepFM.py:64-66
struct CsvField { const char* p; size_t n; };
// splits one line (no trailing newline) into fields; supports "quoted, fields" with "" escapes (unescaped into `scratch`)
void split_csv_line(const char* p, const char* end, std::vector<CsvField>& out, std::string& scratch) {
    out.clear();
    scratch.clear();
    scratch.reserve((size_t)(end - p) + 1);                     // pointers into scratch stay valid
    while (true) {
        if (p < end && *p == '"') {
            const size_t start = scratch.size();
            ++p;
            while (p < end) {
                if (*p == '"') {
                    if (p + 1 < end && p[1] == '"') { scratch.push_back('"'); p += 2; continue; }
                    ++p;
                    break;
                }
                scratch.push_back(*p++);
            }
            out.push_back(CsvField{scratch.data() + start, scratch.size() - start});
            while (p < end && *p != ',') ++p;
        } else {
            const char* q = p;
            while (q < end && *q != ',') ++q;
            out.push_back(CsvField{p, (size_t)(q - p)});
            p = q;
        }
        if (p >= end) break;
        ++p;                                                    // the comma
        if (p == end) { out.push_back(CsvField{p, 0}); break; }
    }
}
bool parse_number(const CsvField& f, double* v) {
    char buf[64];
    if (f.n == 0 || f.n >= sizeof(buf)) return false;
    memcpy(buf, f.p, f.n);
    buf[f.n] = 0;
    char* e = nullptr;
    *v = strtod(buf, &e);
    return e != buf && *e == 0;
}
}  // namespace

int sprk_emb_rank(const float* item_emb, const uint8_t* item_has, int32_t n_items, int32_t D, int32_t item_stride,
                  const float* query_emb, const uint8_t* query_has, int32_t n_queries, int32_t query_stride,
                  const int32_t* cand, int32_t C, double* scores, int32_t* order, void* stream) {
    RoctxRange roctx_range_("sprk_emb_rank");
    if (!item_emb || !query_emb || !cand || !scores) return fail(SPRK_EINVAL, "emb_rank: NULL table / queries / candidates / scores");
    if (n_items < 0 || n_queries < 0 || C < 0 || D < 1 || D > 1024 || item_stride < D || query_stride < D)
        return fail(SPRK_EINVAL, "emb_rank: bad sizes (need 1 <= D <= 1024, strides >= D)");
    if (order && C > ER_MAX_SORT) return fail(SPRK_EINVAL, "emb_rank: ranking supports at most 4096 candidates per query");
    if (n_queries == 0 || C == 0) return SPRK_OK;
    if (order && C <= 1024 && !getenv("SPRK_EMB_RANK_GENERIC")) {   // one wave per query, bitonic network in registers
        const int grid = (n_queries + ERW_WAVES - 1) / ERW_WAVES;
        const int E = C <= 256 ? 4 : 16;
        const size_t lds_w = (size_t)ERW_WAVES * 64 * E * 8 + (size_t)ERW_WAVES * D * 4;
        if (E == 4)
            hipLaunchKernelGGL(k_emb_rank_wave<4>, dim3(grid), dim3(ERW_WAVES * 64), lds_w, (hipStream_t)stream, item_emb, item_has,
                               n_items, D, item_stride, query_emb, query_has, n_queries, query_stride, cand, C, scores, order);
        else
            hipLaunchKernelGGL(k_emb_rank_wave<16>, dim3(grid), dim3(ERW_WAVES * 64), lds_w, (hipStream_t)stream, item_emb, item_has,
                               n_items, D, item_stride, query_emb, query_has, n_queries, query_stride, cand, C, scores, order);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    int P = 0;
    if (order) { P = 2; while (P < C) P <<= 1; }
    const size_t lds = (size_t)P * 12 + (size_t)D * 4 + 16;
    hipLaunchKernelGGL(k_emb_rank, dim3(n_queries), dim3(ER_THREADS), lds, (hipStream_t)stream, item_emb, item_has, n_items, D,
                       item_stride, query_emb, query_has, query_stride, cand, C, P, scores, order);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

}  // extern "C" (helpers below are C++)

namespace {
struct CsvLayout {
    size_t n_cols;
    std::vector<int> id_pos, dense_pos;
};
struct CsvChunkResult {
    std::vector<int32_t> ids;
    std::vector<float> dense;
    int32_t rows = 0;
    int rc = SPRK_OK;                 // first error of the chunk, raised after `rows` good rows
    std::string msg;
};
// Rows of [p, end) appended to `out` (at most max_rows); stops at the first bad value (out.rc / out.msg).
void pack_csv_rows(const char* p, const char* end, const CsvLayout& L, const sprk_csv_col* id_cols, int n_id, const char* const* dense_names,
                   int n_dense, int32_t max_rows, CsvChunkResult& out) {
    std::vector<CsvField> fields;
    std::string scratch;
    char buf[256];
    auto line_end = [&](const char* s) { const void* q = memchr(s, '\n', (size_t)(end - s)); return q ? (const char*)q : end; };
    while (p < end && out.rows < max_rows) {
        const char* le = line_end(p);
        const char* re = (le > p && le[-1] == '\r') ? le - 1 : le;
        if (re > p) {
            split_csv_line(p, re, fields, scratch);
            if (fields.size() == L.n_cols) {                    // ignore_errors=True: other rows are dropped
                const size_t i0 = out.ids.size(), d0 = out.dense.size();
                out.ids.resize(i0 + n_id);
                out.dense.resize(d0 + n_dense);
                for (int j = 0; j < n_id; ++j) {
                    const CsvField& f = fields[L.id_pos[j]];
                    int32_t v;
                    if (id_cols[j].kind == 1) {
                        v = -1;
                        for (int g = 0; g < 19; ++g)
                            if (f.n == strlen(kGenreVocab[g]) && memcmp(f.p, kGenreVocab[g], f.n) == 0) { v = g; break; }
                        if (v >= id_cols[j].vocab) v = -1;
                    } else {
                        double d = 0.0;
                        if (f.n != 0 && !parse_number(f, &d)) {
                            snprintf(buf, sizeof(buf), "%s is not a number", id_cols[j].name);
                            out.rc = SPRK_EINVAL; out.msg = buf; out.ids.resize(i0); out.dense.resize(d0);
                            return;
                        }
                        const long long iv = (long long)d;      // int(float(v)) of the Python packer
                        if (iv < 0 || iv >= id_cols[j].vocab) {
                            snprintf(buf, sizeof(buf), "%s id %lld outside [0, %d) (reference: assert_less_than_num_buckets)", id_cols[j].name, iv,
                                     id_cols[j].vocab);
                            out.rc = SPRK_ERANGE; out.msg = buf; out.ids.resize(i0); out.dense.resize(d0);
                            return;
                        }
                        v = (int32_t)iv;
                    }
                    out.ids[i0 + j] = v;
                }
                for (int j = 0; j < n_dense; ++j) {
                    const CsvField& f = fields[L.dense_pos[j]];
                    double d = 0.0;
                    if (f.n != 0 && !parse_number(f, &d)) {
                        snprintf(buf, sizeof(buf), "%s is not a number", dense_names[j]);
                        out.rc = SPRK_EINVAL; out.msg = buf; out.ids.resize(i0); out.dense.resize(d0);
                        return;
                    }
                    out.dense[d0 + j] = (float)d;
                }
                ++out.rows;
            }
        }
        p = le < end ? le + 1 : end;
    }
}
}  // namespace

extern "C" {

int sprk_pack_csv_mt(const char* text, size_t len, const sprk_csv_col* id_cols, int32_t n_id, const char* const* dense_names,
                     int32_t n_dense, int32_t max_rows, int32_t n_threads, int32_t* ids_out, float* dense_out, int32_t* rows_out) {
    if (!text || !rows_out || n_id < 0 || n_dense < 0 || max_rows < 0) return fail(SPRK_EINVAL, "bad pack_csv arguments");
    if ((n_id > 0 && (!id_cols || !ids_out)) || (n_dense > 0 && (!dense_names || !dense_out))) return fail(SPRK_EINVAL, "NULL column list / output");
    *rows_out = 0;
    const char* p = text;
    const char* const end = text + len;
    // header
    CsvLayout L;
    {
        const void* q = memchr(p, '\n', len);
        const char* le = q ? (const char*)q : end;
        const char* he = (le > p && le[-1] == '\r') ? le - 1 : le;
        std::vector<CsvField> fields;
        std::string scratch;
        split_csv_line(p, he, fields, scratch);
        L.n_cols = fields.size();
        L.id_pos.assign(n_id, -1);
        L.dense_pos.assign(n_dense, -1);
        auto find = [&](const char* name) {
            const size_t n = strlen(name);
            for (size_t c = 0; c < L.n_cols; ++c) if (fields[c].n == n && memcmp(fields[c].p, name, n) == 0) return (int)c;
            return -1;
        };
        for (int j = 0; j < n_id; ++j) if ((L.id_pos[j] = find(id_cols[j].name)) < 0) return fail(SPRK_EINVAL, "CSV has no column %s", id_cols[j].name);
        for (int j = 0; j < n_dense; ++j) if ((L.dense_pos[j] = find(dense_names[j])) < 0) return fail(SPRK_EINVAL, "CSV has no column %s", dense_names[j]);
        p = le < end ? le + 1 : end;
    }
    // chunks of whole lines, one per thread (a text below 1 MiB is not worth a thread start)
    int T = n_threads < 1 ? 1 : (n_threads > 256 ? 256 : n_threads);
    const size_t body = (size_t)(end - p);
    if (body < ((size_t)1 << 20)) T = 1;
    std::vector<const char*> cut(T + 1, end);
    cut[0] = p;
    for (int t = 1; t < T; ++t) {
        const char* c = p + body / T * t;
        if (c < cut[t - 1]) c = cut[t - 1];
        const void* q = c < end ? memchr(c, '\n', (size_t)(end - c)) : nullptr;
        cut[t] = q ? (const char*)q + 1 : end;
    }
    std::vector<CsvChunkResult> res(T);
    if (T == 1) {
        pack_csv_rows(cut[0], cut[1], L, id_cols, n_id, dense_names, n_dense, max_rows, res[0]);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() { pack_csv_rows(cut[t], cut[t + 1], L, id_cols, n_id, dense_names, n_dense, max_rows, res[t]); });
        for (auto& x : th) x.join();
    }
    // stitch in file order: exactly what one pass would have produced (rows beyond max_rows are never looked at)
    int32_t rows = 0;
    for (int t = 0; t < T && rows < max_rows; ++t) {
        const CsvChunkResult& r = res[t];
        const int32_t take = r.rows < max_rows - rows ? r.rows : max_rows - rows;
        if (take > 0) {
            if (n_id) memcpy(ids_out + (size_t)rows * n_id, r.ids.data(), (size_t)take * n_id * sizeof(int32_t));
            if (n_dense) memcpy(dense_out + (size_t)rows * n_dense, r.dense.data(), (size_t)take * n_dense * sizeof(float));
        }
        rows += take;
        if (r.rc != SPRK_OK && rows < max_rows) return fail(r.rc, "row %d: %s", rows, r.msg.c_str());
    }
    *rows_out = rows;
    return SPRK_OK;
}

int sprk_pack_csv(const char* text, size_t len, const sprk_csv_col* id_cols, int32_t n_id, const char* const* dense_names,
                  int32_t n_dense, int32_t max_rows, int32_t* ids_out, float* dense_out, int32_t* rows_out) {
    return sprk_pack_csv_mt(text, len, id_cols, n_id, dense_names, n_dense, max_rows, 1, ids_out, dense_out, rows_out);
}

int sprk_cross_hash(const int32_t* a, const int32_t* b, int32_t B, int64_t num_buckets, int64_t* out, void* stream) {
    if (!a || !b || !out) return fail(SPRK_EINVAL, "NULL argument");
    if (num_buckets <= 0) return fail(SPRK_EINVAL, "num_buckets must be positive");
    if (B < 0) return fail(SPRK_EINVAL, "negative batch");
    if (B == 0) return SPRK_OK;
    int blocks = (B + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_cross_hash, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, B, (unsigned long long)num_buckets, (long long*)out);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

}  // extern "C"

// ---- device ingest: the CSV text is already in HBM (k_csv_pack.h) ----
namespace {
struct DevScratch {
    void* p = nullptr;
    size_t cap = 0;
    int dev = -1;                         // the device `p` lives on: a thread that switches devices gets a new scratch there
    int ensure(size_t bytes) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return fail(SPRK_EHIP, "hipGetDevice"); }
        if (cur == dev && bytes <= cap) return SPRK_OK;
        if (p) (void)hipFree(p);          // hipFree takes a pointer of any device
        p = nullptr; cap = 0; dev = cur;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return fail(SPRK_EHIP, "device scratch of %zu bytes for the CSV tokenizer", want); }
        cap = want;
        return SPRK_OK;
    }
};
thread_local DevScratch g_csv_scratch;
thread_local int g_csv_last_path = -1;

// exclusive scan of n unsigned counters (in -> out, in place allowed), grand total -> *total_dev; sums = scratch of ceil(n / SCAN_TILE)
void scan_u32(const unsigned* in, unsigned* out, size_t n, unsigned* sums, unsigned* total_dev, hipStream_t st) {
    const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, sums, nb, total_dev);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, st, in, n, (const unsigned*)sums, out);
}
}  // namespace

extern "C" {

int sprk_pack_csv_device(const char* text_dev, size_t len, const sprk_csv_col* id_cols, int32_t n_id, const char* const* dense_names,
                         int32_t n_dense, int32_t max_rows, int32_t* ids_dev, float* dense_dev, int32_t* rows_out, void* stream) {
    RoctxRange roctx_range_("sprk_pack_csv_device");
    if (!text_dev || !rows_out || n_id < 0 || n_dense < 0 || max_rows < 0) return fail(SPRK_EINVAL, "bad pack_csv_device arguments");
    if ((n_id > 0 && (!id_cols || !ids_dev)) || (n_dense > 0 && (!dense_names || !dense_dev))) return fail(SPRK_EINVAL, "NULL column list / output");
    if (n_id > CSV_MAX_OUT || n_dense > CSV_MAX_OUT) return fail(SPRK_EINVAL, "the device tokenizer packs at most %d id and %d dense columns", CSV_MAX_OUT, CSV_MAX_OUT);
    if ((uintptr_t)text_dev & 15) return fail(SPRK_EINVAL, "the CSV text must start on a 16-byte boundary in device memory");
    if (len >= ((size_t)1 << 44)) return fail(SPRK_EINVAL, "CSV text too large");
    *rows_out = 0;
    g_csv_last_path = -1;
    if (len == 0) return SPRK_OK;
    hipStream_t st = (hipStream_t)stream;
    // header: the first line comes back to the host and goes through the host tokenizer's own field splitter
    std::vector<char> head(len < 16384 ? len : 16384);
    HIP_TRY(hipMemcpyAsync(head.data(), text_dev, head.size(), hipMemcpyDeviceToHost, st));
    char last = 0;
    HIP_TRY(hipMemcpyAsync(&last, text_dev + len - 1, 1, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const void* q = memchr(head.data(), '\n', head.size());
    if (!q && head.size() < len) return fail(SPRK_EINVAL, "CSV header line longer than %zu bytes", head.size());
    const char* le = q ? (const char*)q : head.data() + head.size();
    const char* he = (le > head.data() && le[-1] == '\r') ? le - 1 : le;
    std::vector<CsvField> fields;
    std::string scratch;
    split_csv_line(head.data(), he, fields, scratch);
    if (fields.size() > CSV_MAX_COLS) return fail(SPRK_EINVAL, "the device tokenizer reads at most %d CSV columns (header has %zu)", CSV_MAX_COLS, fields.size());
    CsvDev L;
    memset(&L, 0, sizeof(L));
    L.n_cols = (int)fields.size(); L.n_id = n_id; L.n_dense = n_dense;
    for (int c = 0; c < CSV_MAX_COLS; ++c) { L.id_head[c] = -1; L.dense_head[c] = -1; }
    auto find = [&](const char* name) {
        const size_t n = strlen(name);
        for (size_t c = 0; c < fields.size(); ++c) if (fields[c].n == n && memcmp(fields[c].p, name, n) == 0) return (int)c;
        return -1;
    };
    for (int j = n_id - 1; j >= 0; --j) {                          // (back to front: every column's list ends up in output order)
        const int c = find(id_cols[j].name);
        if (c < 0) return fail(SPRK_EINVAL, "CSV has no column %s", id_cols[j].name);
        L.id_next[j] = L.id_head[c]; L.id_head[c] = (short)j;
        L.id_kind[j] = id_cols[j].kind; L.id_vocab[j] = id_cols[j].vocab;
    }
    for (int j = n_dense - 1; j >= 0; --j) {
        const int c = find(dense_names[j]);
        if (c < 0) return fail(SPRK_EINVAL, "CSV has no column %s", dense_names[j]);
        L.dense_next[j] = L.dense_head[c]; L.dense_head[c] = (short)j;
    }
    for (int c = 0; c < L.n_cols; ++c) {
        for (int j = L.id_head[c]; j >= 0; j = L.id_next[j]) L.role[c] |= L.id_kind[j] == 1 ? 2 : 1;
        if (L.dense_head[c] >= 0) L.role[c] |= 1;
    }
    {
        // perfect hash of the 19 genre strings into 32 slots: the first odd multiplier without a collision
        unsigned long long lo[19], hi[19];
        unsigned len[19];
        for (int g = 0; g < 19; ++g) {
            const size_t n = strlen(kGenreVocab[g]);
            lo[g] = hi[g] = 0;
            len[g] = (unsigned)n;
            for (size_t k = 0; k < n && k < 16; ++k) (k < 8 ? lo[g] : hi[g]) |= (unsigned long long)(unsigned char)kGenreVocab[g][k] << (8 * (k & 7));
        }
        unsigned long long mul = 0x9E3779B97F4A7C15ull;
        for (int tries = 0; tries < 100000; ++tries, mul += 0x632BE59BD9B4E019ull * 2) {
            unsigned used = 0;
            bool ok = true;
            for (int g = 0; g < 19 && ok; ++g) {
                const unsigned sl = csv_genre_slot(lo[g], hi[g], len[g], mul | 1);
                ok = !(used & (1u << sl));
                used |= 1u << sl;
            }
            if (ok) break;
        }
        L.g_mul = mul | 1;
        for (int sl = 0; sl < 32; ++sl) { L.gt_idx[sl] = -1; L.gt_len[sl] = -1; }
        for (int g = 0; g < 19; ++g) {
            const unsigned sl = csv_genre_slot(lo[g], hi[g], len[g], L.g_mul);
            if (L.gt_idx[sl] >= 0) return fail(SPRK_EINVAL, "no perfect hash for the genre vocabulary");
            L.gt_lo[sl] = lo[g]; L.gt_hi[sl] = hi[g]; L.gt_len[sl] = (signed char)len[g]; L.gt_idx[sl] = (signed char)g;
        }
    }
    // pass 1: newlines per chunk
    const size_t n_chunks = (len + CSV_CHUNK - 1) / CSV_CHUNK;
    const size_t sums_a = (n_chunks + SCAN_TILE - 1) / SCAN_TILE;
    const size_t fixed = 256 + sizeof(CsvErr) * 64;                 // flags | totals | first_err | n_errs, then the error records
    size_t need = fixed + (n_chunks + sums_a + 64) * sizeof(unsigned);
    if (int rc = g_csv_scratch.ensure(need)) return rc;
    auto carve = [&]() { return (char*)g_csv_scratch.p; };
    unsigned* totals = (unsigned*)(carve() + 16);                   // [0] newlines, [1] kept lines
    unsigned long long* first_err = (unsigned long long*)(carve() + 32);
    unsigned* n_errs = (unsigned*)(carve() + 48);
    CsvErr* errs = (CsvErr*)(carve() + 256);
    unsigned* counts = (unsigned*)(carve() + fixed);
    unsigned* sums = counts + n_chunks;
    HIP_TRY(hipMemsetAsync(carve(), 0, 256, st));
    HIP_TRY(hipMemsetAsync(first_err, 0xFF, sizeof(unsigned long long), st));
    const unsigned char* text = (const unsigned char*)text_dev;
    hipLaunchKernelGGL(k_csv_count, dim3((unsigned)n_chunks), dim3(256), 0, st, text, len, counts);
    scan_u32(counts, counts, n_chunks, sums, totals, st);
    unsigned* drops = (unsigned*)(carve() + 52);
    unsigned h_nl = 0, h_kept = 0, h_nerr = 0, h_drops = 0;
    unsigned long long h_first = ~0ull;
    const char* two = getenv("SPRK_CSV_TWO_PASS");              // A/B switch: "1" = always the exact keep -> scan -> parse sequence
    bool exact = two && two[0] == '1';
    HIP_TRY(hipMemcpyAsync(&h_nl, totals, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const size_t n_lines = (size_t)h_nl + (last != '\n' ? 1 : 0);
    if (n_lines >= ((size_t)1 << 31)) return fail(SPRK_EINVAL, "more than 2^31 lines");
    if (n_lines <= 1) return SPRK_OK;                              // header only
    // passes 2-4 need nl[] and keep[] / pos[]: grow the scratch (contents so far are carried over by redoing pass 1's scan)
    const size_t sums_b = (n_lines + SCAN_TILE - 1) / SCAN_TILE;
    const size_t off_nl = (fixed + (n_chunks + sums_a + 64) * sizeof(unsigned) + 255) & ~(size_t)255;
    const size_t off_keep = off_nl + ((size_t)h_nl + 1) * sizeof(unsigned long long);
    need = off_keep + (2 * n_lines + sums_b + 64) * sizeof(unsigned);
    if (need > g_csv_scratch.cap) {
        // (first call on a text of this size: allocate the full scratch and run pass 1 again into it)
        if (int rc = g_csv_scratch.ensure(need)) return rc;
        totals = (unsigned*)(carve() + 16); first_err = (unsigned long long*)(carve() + 32);
        n_errs = (unsigned*)(carve() + 48); errs = (CsvErr*)(carve() + 256); counts = (unsigned*)(carve() + fixed); sums = counts + n_chunks;
        HIP_TRY(hipMemsetAsync(carve(), 0, 256, st));
        HIP_TRY(hipMemsetAsync(first_err, 0xFF, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_csv_count, dim3((unsigned)n_chunks), dim3(256), 0, st, text, len, counts);
        scan_u32(counts, counts, n_chunks, sums, totals, st);
    }
    unsigned long long* nl = (unsigned long long*)(carve() + off_nl);
    unsigned* keep = (unsigned*)(carve() + off_keep);
    unsigned* pos = keep + n_lines;
    unsigned* sums2 = pos + n_lines;
    drops = (unsigned*)(carve() + 52);
    hipLaunchKernelGGL(k_csv_mark, dim3((unsigned)n_chunks), dim3(256), 0, st, text, len, (const unsigned*)counts, nl);
    const unsigned lb = (unsigned)((n_lines + 255) / 256);
    // LDS piece per workgroup of 256 lines: twice the average, so that more workgroups share a CU when lines are short
    size_t cap = (2 * 256 * (len / n_lines + 1) + 4095) & ~(size_t)4095;
    if (cap < 8192) cap = 8192;
    if (cap > 48 * 1024) cap = 48 * 1024;
    const unsigned lds_cap = (unsigned)cap;
    if (!exact) {
        // optimistic pass over the lines that can hold the first max_rows rows if none is dropped
        const size_t lines_opt = n_lines - 1 <= (size_t)max_rows ? n_lines : (size_t)max_rows + 1;
        const unsigned lbo = (unsigned)((lines_opt + 255) / 256);
        if (lines_opt > 1)
            hipLaunchKernelGGL(k_csv_parse<true>, dim3(lbo), dim3(256), lds_cap + CSV_LDS_SLACK + CSV_LDS_GENRE, st, L, text, len, (const unsigned long long*)nl, h_nl,
                               (unsigned)lines_opt, (const unsigned*)nullptr, (const unsigned*)nullptr, (unsigned)max_rows, lds_cap, ids_dev, dense_dev,
                               first_err, errs, n_errs, drops);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h_drops, drops, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&h_first, first_err, sizeof(h_first), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&h_nerr, n_errs, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        h_kept = (unsigned)(lines_opt - 1);
        g_csv_last_path = 1;
        if (h_drops) {                                             // some line is not a row: its successors are misplaced
            exact = true;
            HIP_TRY(hipMemsetAsync(first_err, 0xFF, sizeof(unsigned long long), st));
            HIP_TRY(hipMemsetAsync(n_errs, 0, sizeof(unsigned), st));
        }
    }
    if (exact) {
        g_csv_last_path = 2;
        hipLaunchKernelGGL(k_csv_keep, dim3(lb), dim3(256), lds_cap + CSV_LDS_SLACK + CSV_LDS_GENRE, st, text, len, (const unsigned long long*)nl, h_nl, (unsigned)n_lines,
                           L.n_cols, lds_cap, keep);
        scan_u32(keep, pos, n_lines, sums2, totals + 1, st);
        hipLaunchKernelGGL(k_csv_parse<false>, dim3(lb), dim3(256), lds_cap + CSV_LDS_SLACK + CSV_LDS_GENRE, st, L, text, len, (const unsigned long long*)nl, h_nl,
                           (unsigned)n_lines, (const unsigned*)keep, (const unsigned*)pos, (unsigned)max_rows, lds_cap, ids_dev, dense_dev, first_err, errs,
                           n_errs, drops);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h_kept, totals + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&h_first, first_err, sizeof(h_first), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&h_nerr, n_errs, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (h_first != ~0ull) {
        std::vector<CsvErr> rec(h_nerr < 64 ? h_nerr : 64);
        if (!rec.empty()) HIP_TRY(hipMemcpy(rec.data(), errs, rec.size() * sizeof(CsvErr), hipMemcpyDeviceToHost));
        const unsigned row = (unsigned)(h_first >> 20);
        const int code = (int)(h_first & 15);
        const CsvErr* hit = nullptr;
        for (const CsvErr& e : rec) if (e.key == h_first) { hit = &e; break; }
        const char* name = !hit ? "a column" : (hit->is_dense ? dense_names[hit->out_col] : id_cols[hit->out_col].name);
        if (code == 1) {
            if (hit) return fail(SPRK_ERANGE, "row %u: %s id %lld outside [0, %d) (reference: assert_less_than_num_buckets)", row, name, hit->value, id_cols[hit->out_col].vocab);
            return fail(SPRK_ERANGE, "row %u: an identity id is outside its bucket range (reference: assert_less_than_num_buckets)", row);
        }
        return fail(SPRK_EKIND, "row %u: %s holds a value the device tokenizer does not convert exactly (not a plain decimal of at most 15 digits): use sprk_pack_csv", row, name);
    }
    *rows_out = (int32_t)(h_kept < (unsigned)max_rows ? h_kept : (unsigned)max_rows);
    return SPRK_OK;
}

int sprk_csv_last_path(void) { return g_csv_last_path; }

}  // extern "C"

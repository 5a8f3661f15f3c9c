// k_csv_pack.h -- CSV text -> the packed ids [rows][n_id] int32 / dense [rows][n_dense] float32 arrays ON THE DEVICE
// ("GPU-side tokenizer"): the device twin of sprk_pack_csv (same rules, same bits), i.e. of
// the reference's get_dataset = tf.data.experimental.make_csv_dataset(..., na_value="0", ignore_errors=True) (DeepFM.py:14-22)
// plus the feature columns' id resolution (DeepFM.py:54-76).  Included inside sparrow_hip.hip's anonymous namespace.
//
// Byte work, bound by HBM (the text is read, nothing is computed): four passes over line-sized pieces, all integer.
//   k_csv_count      newlines per 4-KB chunk (16-byte loads)
//   (scan)           chunk counts -> first line index of every chunk
//   k_csv_mark       byte offset of every newline -> nl[]           (line i = (nl[i-1], nl[i]), nl[-1] = -1)
//   k_csv_parse<OPT> the common case in ONE pass: line i is output row i - 1, lines that should have been dropped are counted and,
//                    if there were any, the exact sequence below runs instead
//   k_csv_keep       one thread per line: '\r' stripped, empty lines and lines whose field count differs from the header's are
//                    dropped (ignore_errors=True)                   -> keep[]
//                    (fields are split exactly as the host tokenizer splits them: a field that STARTS with '"' runs to its
//                    closing quote, "" inside it is an escaped quote, whatever follows up to the next comma is ignored; the
//                    reference's Spark-written sample files spell an empty string "")
//   (scan)           keep[] -> output row of every kept line
//   k_csv_parse      one thread per kept line (output row < max_rows): fields in order, the ones a column list names are
//                    converted -- identity ids: empty -> 0, decimal -> (long long) truncation, range check; genre strings ->
//                    position in the 19-entry vocabulary or -1; numerics: empty -> 0.0, decimal -> float
// Decimal -> double is the exact fast path of strtod (Clinger): up to 15 significant digits m and |e10| <= 22 give m * 10^e10 or
// m / 10^-e10 in ONE correctly rounded IEEE operation, which is what strtod returns; the host tokenizer then narrows with
// (float), and so does this one -- bit-identical for every field of that shape.  A field outside it (more digits, larger
// exponents, "inf", hex, blanks, text in a numeric column, an escaped quote inside a quoted number) marks its row HARD: the
// call fails and names the row, it never guesses (the host tokenizer parses such files).  The first bad row in file order decides the error, as on the host.

#define CSV_MAX_COLS 128
#define CSV_MAX_OUT 64
#define CSV_CHUNK 4096                        // bytes per workgroup in the newline passes: 256 threads x 16 bytes

struct CsvDev {
    int n_cols, n_id, n_dense;
    short id_head[CSV_MAX_COLS], dense_head[CSV_MAX_COLS];   // first output fed by CSV column c (-1: none)
    short id_next[CSV_MAX_OUT], dense_next[CSV_MAX_OUT];     // next output fed by the same column (-1: end)
    int id_kind[CSV_MAX_OUT], id_vocab[CSV_MAX_OUT];
    unsigned char role[CSV_MAX_COLS];                        // bit 0: column c feeds a numeric output, bit 1: a genre output
    // the genre vocabulary (DeepFM.py:64-66) behind a perfect hash: slot = (key * g_mul) >> 59 with key = lo ^ hi * C ^ len; a slot
    // holds the entry's 16 little-endian bytes, its length and its vocabulary position (-1: empty)
    unsigned long long g_mul;
    unsigned long long gt_lo[32], gt_hi[32];
    signed char gt_len[32], gt_idx[32];
};
#define CSV_GENRE_C 0x9E3779B97F4A7C15ull
__host__ __device__ inline unsigned csv_genre_slot(unsigned long long lo, unsigned long long hi, unsigned len, unsigned long long mul) {
    return (unsigned)((((lo ^ (hi * CSV_GENRE_C)) ^ len) * mul) >> 59);
}
// error record of a row that cannot be packed: code 1 = identity id outside [0, vocab), 2 = HARD (see above)
struct CsvErr { unsigned long long key; int code, out_col, is_dense; long long value; };

static __device__ __constant__ double kCsvP10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                              1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// newlines among the 16 bytes at t + i (bytes at or beyond len do not count)
__device__ __forceinline__ unsigned csv_nl_mask(const unsigned char* __restrict__ t, size_t i, size_t len) {
    unsigned m = 0;
    if (i + 16 <= len) {
        const uint4 w = *reinterpret_cast<const uint4*>(t + i);
        const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned c = (ww[k] >> (8 * b)) & 0xFF;
                m |= (c == '\n') ? 1u << (4 * k + b) : 0u;
            }
    } else {
        for (int b = 0; b < 16 && i + b < len; ++b) {
            const unsigned c = t[i + b];
            m |= (c == '\n') ? 1u << b : 0u;
        }
    }
    return m;
}

static __global__ __launch_bounds__(256) void k_csv_count(const unsigned char* __restrict__ text, size_t len, unsigned* __restrict__ counts) {
    const size_t i = (size_t)blockIdx.x * CSV_CHUNK + threadIdx.x * 16;
    unsigned n = i < len ? __popc(csv_nl_mask(text, i, len)) : 0;
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

static __global__ __launch_bounds__(256) void k_csv_mark(const unsigned char* __restrict__ text, size_t len, const unsigned* __restrict__ first,
                                                  unsigned long long* __restrict__ nl) {
    const size_t i = (size_t)blockIdx.x * CSV_CHUNK + threadIdx.x * 16;
    const unsigned m = i < len ? csv_nl_mask(text, i, len) : 0;
    const unsigned n = __popc(m);
    // exclusive prefix of n over the workgroup's 256 threads
    unsigned incl = n;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if ((int)(threadIdx.x & 63) >= d) incl += o; }
    __shared__ unsigned wsum[4];
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned base = first[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    unsigned k = base + incl - n;
    for (unsigned mm = m; mm; mm &= mm - 1) nl[k++] = i + (unsigned)__builtin_ctz(mm);
}

// Readers: the text in global memory, or a workgroup's 256 lines staged in LDS.  win(i) = the 8 bytes at offsets i .. i+7
// (little endian; bytes past the staged piece / the text are unspecified and never looked at).
// Positions are offsets into the text (global reader: size_t) or into the staged piece (LDS reader: 32 bits -- the walk over a
// line is all position arithmetic, and 64-bit compares / adds cost two VALU instructions each).
struct CsvRdGlobal {
    typedef size_t pos_t;
    const unsigned char* __restrict__ t;
    size_t len;
    __device__ __forceinline__ unsigned operator[](size_t i) const { return t[i]; }
    __device__ __forceinline__ unsigned long long win(size_t i) const {
        unsigned long long w = 0;
        for (int k = 0; k < 8 && i + k < len; ++k) w |= (unsigned long long)t[i + k] << (8 * k);
        return w;
    }
};
struct CsvRdLds {
    typedef unsigned pos_t;
    const unsigned char* lds;
    __device__ __forceinline__ unsigned operator[](unsigned o) const { return lds[o]; }
    __device__ __forceinline__ unsigned long long win(unsigned o) const {
        const unsigned sh = 8 * (o & 7);
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(lds + (o & ~7u));
        const unsigned long long lo = q[0], hi = q[1];             // two aligned 8-byte LDS reads (one ds_read2_b64)
        return (lo >> sh) | ((hi << 1) << (63 - sh));               // branch-free for sh == 0 too
    }
};
#define CSV_LDS_SLACK 32
#define CSV_LDS_GENRE 640                      // bytes behind the staged text: the genre hash table (32 x {lo, hi}, 32 lengths, 32 positions)

// first byte == c among the 8 bytes of w: its index, or 8 (exact for the lowest match: the classic zero-byte test)
__device__ __forceinline__ unsigned csv_find8(unsigned long long w, unsigned c) {
    const unsigned long long x = w ^ (0x0101010101010101ull * c);
    const unsigned long long z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
    return z ? (unsigned)__builtin_ctzll(z) >> 3 : 8u;
}
// position of the next ',' in [p, hi), or hi -- eight bytes per step
template <class R>
__device__ __forceinline__ typename R::pos_t csv_next_comma(const R& t, typename R::pos_t p, typename R::pos_t hi, unsigned long long w) {
    // w = t.win(p), already in hand
    while (p < hi) {
        const unsigned k = csv_find8(w, ',');
        if (k < 8) return p + k < hi ? p + k : hi;
        p += 8;
        if (p < hi) w = t.win(p);
    }
    return hi;
}

// The next field of the line [.., hi) starting at p, split as the host tokenizer's split_csv_line does: content = [a, b),
// esc = the quoted content holds an escaped quote; p is left on the comma that ends the field (or at hi).  A field that STARTS
// with '"' runs to its closing quote ("" inside is an escaped quote), whatever follows up to the next comma is ignored.
template <class R>
__device__ __forceinline__ void csv_field(const R& t, typename R::pos_t hi, typename R::pos_t& p, typename R::pos_t& a,
                                          typename R::pos_t& b, bool& esc) {
    esc = false;
    const unsigned long long w0 = p < hi ? t.win(p) : 0ull;       // ONE read serves the quote test and the first comma search
    if (p < hi && (w0 & 0xFF) == '"') {
        ++p;
        a = p;
        b = hi;                                                   // an unclosed quote runs to the end of the line
        while (p < hi) {
            if (t[p] == '"') {
                if (p + 1 < hi && t[p + 1] == '"') { esc = true; p += 2; continue; }
                b = p;
                ++p;
                break;
            }
            ++p;
        }
        p = csv_next_comma(t, p, hi, p < hi ? t.win(p) : 0ull);
    } else {
        a = p;
        p = csv_next_comma(t, p, hi, w0);
        b = p;
    }
}

// Runs body(reader, i, lo, hi) for the workgroup's 256 lines [256 blockIdx.x, ..) ([lo, hi) = line i without its newline and
// a '\r' before it): their bytes are one contiguous piece of the text, brought into LDS with 16-byte loads when it fits
// lds_cap bytes (one thread per line then walks its line out of LDS instead of issuing 64-address global loads), read in place
// otherwise.
template <class Body>
__device__ __forceinline__ void csv_stage_lines(const unsigned char* __restrict__ text, size_t len, const unsigned long long* __restrict__ nl,
                                                unsigned n_nl, unsigned n_lines, unsigned lds_cap, const CsvDev* L, Body body) {
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
    if (L && threadIdx.x < 32) {
        // the genre hash table next to the text: a per-lane slot lookup is an LDS read, not a dependent load from the kernel
        // arguments (PMC after the instruction diet: 61 % of the wave cycles in s_waitcnt)
        unsigned long long* g = reinterpret_cast<unsigned long long*>(lds + lds_cap + CSV_LDS_SLACK);
        g[2 * threadIdx.x] = L->gt_lo[threadIdx.x];
        g[2 * threadIdx.x + 1] = L->gt_hi[threadIdx.x];
        signed char* gl = reinterpret_cast<signed char*>(g + 64);
        gl[threadIdx.x] = L->gt_len[threadIdx.x];
        gl[32 + threadIdx.x] = L->gt_idx[threadIdx.x];
    }
    const unsigned i0 = blockIdx.x * 256;
    const unsigned i1 = i0 + 256 < n_lines ? i0 + 256 : n_lines;
    const size_t g0 = i0 == 0 ? 0 : (size_t)nl[i0 - 1] + 1;
    const size_t g1 = i1 - 1 < n_nl ? (size_t)nl[i1 - 1] : len;
    const size_t base = g0 & ~(size_t)15;
    const size_t bytes = g1 - base;
    const unsigned i = i0 + threadIdx.x;
    size_t lo = 0, hi = 0;
    if (i < n_lines) {
        lo = i == 0 ? 0 : (size_t)nl[i - 1] + 1;
        hi = i < n_nl ? (size_t)nl[i] : len;
    }
    if (bytes <= lds_cap) {                                       // workgroup-uniform
        for (size_t o = (size_t)threadIdx.x * 16; o < bytes; o += 256 * 16) {
            if (base + o + 16 <= len) {
                *reinterpret_cast<uint4*>(lds + o) = *reinterpret_cast<const uint4*>(text + base + o);
            } else {
                for (int k = 0; k < 16 && base + o + k < len; ++k) lds[o + k] = text[base + o + k];
            }
        }
        __syncthreads();
        if (i >= n_lines) return;
        const CsvRdLds rd{lds};
        unsigned lo32 = (unsigned)(lo - base), hi32 = (unsigned)(hi - base);
        if (hi32 > lo32 && rd[hi32 - 1] == '\r') --hi32;
        body(rd, i, lo32, hi32);
    } else {
        __syncthreads();                                          // (the genre table)
        if (i >= n_lines) return;
        const CsvRdGlobal rd{text, len};
        if (hi > lo && rd[hi - 1] == '\r') --hi;
        body(rd, i, lo, hi);
    }
}

static __global__ __launch_bounds__(256) void k_csv_keep(const unsigned char* __restrict__ text, size_t len, const unsigned long long* __restrict__ nl,
                                                  unsigned n_nl, unsigned n_lines, int n_cols, unsigned lds_cap, unsigned* __restrict__ keep) {
    csv_stage_lines(text, len, nl, n_nl, n_lines, lds_cap, nullptr, [&](auto rd, unsigned i, auto lo, auto hi) {
        unsigned k = 0;
        if (i > 0 && hi > lo) {                                   // line 0 is the header; empty lines are skipped
            int fields = 0;
            decltype(lo) p = lo, a, b;
            bool esc;
            for (;;) {
                csv_field(rd, hi, p, a, b, esc);
                ++fields;
                if (p >= hi) break;
                ++p;                                              // the comma
                if (p == hi) { ++fields; break; }                 // a trailing comma: one more, empty, field
            }
            k = fields == n_cols;
        }
        keep[i] = k;
    });
}

// four ASCII digits, first character in the lowest byte -> their value (no validation)
__device__ __forceinline__ unsigned csv_parse4(unsigned x) {
    x -= 0x30303030u;
    x = x * 10u + (x >> 8);                                       // bytes 0 and 2 now hold d0 d1 and d2 d3 as two-digit numbers
    return (x & 0xFFu) * 100u + ((x >> 16) & 0xFFu);
}
// The common shape of a numeric field without a loop and without a branch per character: at most 8 characters,
// [sign] digits [. digits].  true -> out; false -> anything else (the caller runs the general csv_number).
// PMC on the first version of k_csv_parse: 3 082 SALU + 1 743 VALU instructions per wave of 64 lines, most of them exec-mask
// bookkeeping for csv_number's per-character branches.
template <class R>
__device__ __forceinline__ bool csv_number_short(const R& t, typename R::pos_t a, typename R::pos_t b, double& out) {
    const unsigned n0 = (unsigned)(b - a);                        // 1 .. 8 (checked by the caller)
    unsigned long long w = t.win(a);
    const unsigned c0 = (unsigned)w & 0xFFu;
    const bool sgn = c0 == '-' || c0 == '+';
    w = sgn ? w >> 8 : w;
    const unsigned n = n0 - (sgn ? 1u : 0u);                      // characters after the sign, 0 .. 8
    w &= n >= 8 ? ~0ull : (1ull << (8 * n)) - 1;                  // bytes past the field -> 0
    const unsigned d = csv_find8(w, '.');                         // position of the first '.', 8 = none
    const bool dot = d < 8;
    const unsigned long long below = dot ? (1ull << (8 * d)) - 1 : ~0ull;
    w = dot ? (w & below) | ((w >> 8) & ~below) : w;              // the dot removed: later characters move down one place
    const unsigned nd = n - (dot ? 1u : 0u);                      // digits, in the low nd bytes
    const unsigned frac = dot ? n - 1 - d : 0u;                   // ... of which behind the dot
    // leading '0's in front: the digits end up in the high bytes, first character still in the lower byte
    const unsigned long long al = nd >= 8 ? w : (w << ((8 * (8 - nd)) & 63)) | (0x3030303030303030ull >> (8 * nd));   // (nd = 0 fails below)
    const bool digits = (al & 0xF0F0F0F0F0F0F0F0ull) == 0x3030303030303030ull &&
                        ((al + 0x0606060606060606ull) & 0xF0F0F0F0F0F0F0F0ull) == 0x3030303030303030ull;
    if (!(digits && nd >= 1 && nd <= 8)) return false;
    const unsigned m = csv_parse4((unsigned)al) * 10000u + csv_parse4((unsigned)(al >> 32));
    double v = (double)m;                                         // < 10^8: exact
    if (frac) v = v / kCsvP10[frac];                              // one correctly rounded division, as the general path
    out = c0 == '-' ? -v : v;
    return true;
}

// decimal field [a, b) -> double; returns 0 = value, 1 = empty, 2 = HARD
template <class R>
__device__ __forceinline__ int csv_number(const R& t, typename R::pos_t a, typename R::pos_t b, double& out) {
    out = 0.0;
    if (a == b) return 1;
    unsigned long long w = t.win(a);                              // the field's bytes, eight at a time
    unsigned left = 8;
    typename R::pos_t i = a;
    auto next = [&]() -> unsigned {
        if (left == 0) { w = t.win(i); left = 8; }
        const unsigned c = (unsigned)w & 0xFF;
        w >>= 8; --left; ++i;
        return c;
    };
    unsigned c = next();
    bool neg = false;
    if (c == '-' || c == '+') {
        neg = c == '-';
        if (i >= b) return 2;
        c = next();
    }
    unsigned long long m = 0;
    int nd = 0, e10 = 0;
    bool digit = false, dot = false, hard = false;
    for (;;) {
        if (c >= '0' && c <= '9') {
            digit = true;
            if (m == 0 && c == '0') { if (dot) --e10; }               // leading zeros
            else if (nd < 15) { m = m * 10 + (c - '0'); ++nd; if (dot) --e10; }
            else { hard |= c != '0'; if (!dot) ++e10; }               // a 16th significant digit: exact only if it is a zero
        } else if (c == '.') {
            if (dot) return 2;
            dot = true;
        } else if (c == 'e' || c == 'E') {
            if (!digit || i >= b) return 2;
            c = next();
            bool eneg = false;
            if (c == '-' || c == '+') {
                eneg = c == '-';
                if (i >= b) return 2;
                c = next();
            }
            int ex = 0;
            for (;;) {
                if (c < '0' || c > '9') return 2;
                if (ex < 10000) ex = ex * 10 + (int)(c - '0');
                if (i >= b) break;
                c = next();
            }
            e10 += eneg ? -ex : ex;
            break;
        } else {
            return 2;                                                 // blanks, "inf", "nan", hex, text: the host tokenizer decides
        }
        if (i >= b) break;
        c = next();
    }
    if (!digit || hard) return 2;
    double v;
    if (m == 0) v = 0.0;
    else if (e10 == 0) v = (double)m;
    else if (e10 < 0 && e10 >= -22) v = (double)m / kCsvP10[-e10];
    else if (e10 > 0 && e10 <= 22) v = (double)m * kCsvP10[e10];
    else return 2;
    out = neg ? -v : v;
    return 0;
}

// where the packed values and the error records go
struct CsvSink {
    int* __restrict__ ids;
    float* __restrict__ dense;
    unsigned long long* __restrict__ first_err;
    CsvErr* __restrict__ errs;
    unsigned* __restrict__ n_errs;
    __device__ __forceinline__ void report(unsigned row, int c, int code, int out_col, int is_dense, long long value) const {
        const unsigned long long key = ((unsigned long long)row << 20) | ((unsigned long long)(unsigned)c << 4) | (unsigned)code;
        atomicMin(first_err, key);
        const unsigned s = atomicAdd(n_errs, 1u);
        if (s < 64) { errs[s].key = key; errs[s].code = code; errs[s].out_col = out_col; errs[s].is_dense = is_dense; errs[s].value = value; }
    }
};
// Field c = [a, b) of output row `row` (esc: its quoted content holds an escaped quote): converted for every output the column
// lists name.  T = the layout (kernel argument, or its copy in LDS when c differs per lane), g_tab / g_len = the genre hash table.
template <class R, class P>
__device__ __forceinline__ void csv_emit(const CsvDev& T, const unsigned long long* g_tab, const signed char* g_len, const R& rd, P a, P b,
                                         bool esc, int role, int c, unsigned row, const CsvSink& out) {
    double v = 0.0;
    int st = 1;
    if (role & 1) {
        if (esc) st = 2;
        else if (a == b) st = 1;
        else if ((unsigned)(b - a) <= 8 && csv_number_short(rd, a, b, v)) st = 0;
        else st = csv_number(rd, a, b, v);
    }
    unsigned long long w0 = 0, w1 = 0;
    const unsigned gn = (unsigned)(b - a);
    if ((role & 2) && gn >= 1 && gn <= 16) {                   // the field's bytes as two little-endian words
        w0 = rd.win(a);
        if (gn < 8) w0 &= ~0ull >> (64 - 8 * gn);
        if (gn > 8) { w1 = rd.win(a + 8); if (gn < 16) w1 &= ~0ull >> (64 - 8 * (gn - 8)); }
    }
    for (int o = T.id_head[c]; o >= 0; o = T.id_next[o]) {
        int val;
        if (T.id_kind[o] == 1) {                                  // genre vocabulary (exact match) or -1
            val = -1;
            if (gn >= 1 && gn <= 16) {
                const unsigned sl = csv_genre_slot(w0, w1, gn, T.g_mul);
                if (g_len[sl] == (int)gn && g_tab[2 * sl] == w0 && g_tab[2 * sl + 1] == w1) val = g_len[32 + sl];
            }
            if (val >= T.id_vocab[o]) val = -1;
        } else if (st == 2) {
            out.report(row, c, 2, o, 0, 0);
            val = 0;
        } else {
            const long long iv = (long long)v;                    // int(float(v)) of the Python packer; empty -> 0
            if (iv < 0 || iv >= T.id_vocab[o]) out.report(row, c, 1, o, 0, iv);
            val = (int)iv;
        }
        out.ids[(size_t)row * T.n_id + o] = val;
    }
    for (int o = T.dense_head[c]; o >= 0; o = T.dense_next[o]) {
        if (st == 2) out.report(row, c, 2, o, 1, 0);
        out.dense[(size_t)row * T.n_dense + o] = (float)v;
    }
}

// OPT: the optimistic single pass -- every line i >= 1 is taken as output row i - 1; lines that would have been dropped (empty,
// wrong field count) are counted in *drops and the host then reruns the exact keep -> scan -> parse sequence.  Sample files
// have no such lines, so the common case reads the text one time less.
template <bool OPT>
__global__ __launch_bounds__(256) void k_csv_parse(const CsvDev L, const unsigned char* __restrict__ text, size_t len,
                                                   const unsigned long long* __restrict__ nl, unsigned n_nl, unsigned n_lines,
                                                   const unsigned* __restrict__ keep, const unsigned* __restrict__ pos, unsigned max_rows,
                                                   unsigned lds_cap, int* __restrict__ ids, float* __restrict__ dense,
                                                   unsigned long long* __restrict__ first_err, CsvErr* __restrict__ errs,
                                                   unsigned* __restrict__ n_errs, unsigned* __restrict__ drops) {
    const unsigned long long* g_tab = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned char*>(smem) + lds_cap + CSV_LDS_SLACK);
    const signed char* g_len = reinterpret_cast<const signed char*>(g_tab + 64);
    csv_stage_lines(text, len, nl, n_nl, n_lines, lds_cap, &L, [&](auto rd, unsigned i, auto lo, auto hi) {
        unsigned row;
        if constexpr (OPT) {
            if (i == 0) return;
            if (hi == lo) { atomicAdd(drops, 1u); return; }
            row = i - 1;
        } else {
            if (!keep[i]) return;
            row = pos[i];
        }
        if (row >= max_rows) return;
        const CsvSink sink{ids, dense, first_err, errs, n_errs};
        // Every kept line has exactly n_cols fields, so the field loop runs a uniform n_cols times (the column's role, its output
        // lists: scalar); a lane whose line ends early (OPT only: such a line is dropped) idles through the rest.
        decltype(lo) p = lo;
        bool ended = false, short_line = false;
        for (int c = 0; c < L.n_cols; ++c) {
            if (ended) { short_line = true; continue; }
            decltype(lo) a, b;
            bool esc;
            csv_field(rd, hi, p, a, b, esc);                      // (p == hi on entry: the empty field after a trailing comma)
            if (p >= hi) ended = true;
            else ++p;                                             // the comma
            const int role = L.role[c];
            if (role) csv_emit(L, g_tab, g_len, rd, a, b, esc, role, c, row, sink);
        }
        if constexpr (OPT) {
            if (short_line || !ended) atomicAdd(drops, 1u);       // fewer or more fields than the header: not a row
        }
    });
}

// ---- exclusive scan of unsigned counters (three small kernels; totals of a few million elements) ----
#define SCAN_TILE 2048                        // elements per workgroup: 256 threads x 8
static __global__ __launch_bounds__(256) void k_scan_tiles(const unsigned* __restrict__ in, size_t n, unsigned* __restrict__ sums) {
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += base + k < n ? in[base + k] : 0u;
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// one workgroup: sums[0 .. nb) -> exclusive prefix in place, grand total -> *total
static __global__ __launch_bounds__(256) void k_scan_sums(unsigned* __restrict__ sums, size_t nb, unsigned* __restrict__ total) {
    __shared__ unsigned wsum[4];
    __shared__ unsigned carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t b0 = 0; b0 < nb; b0 += 256) {
        const size_t i = b0 + threadIdx.x;
        const unsigned v = i < nb ? sums[i] : 0u;
        unsigned incl = v;
        for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if ((int)(threadIdx.x & 63) >= d) incl += o; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned base = carry_s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
        if (i < nb) sums[i] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
static __global__ __launch_bounds__(256) void k_scan_apply(const unsigned* __restrict__ in, size_t n, const unsigned* __restrict__ sums,
                                                    unsigned* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    unsigned v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    unsigned incl = s;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if ((int)(threadIdx.x & 63) >= d) incl += o; }
    __shared__ unsigned wsum[4];
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned run = sums[blockIdx.x] + incl - s;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wsum[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

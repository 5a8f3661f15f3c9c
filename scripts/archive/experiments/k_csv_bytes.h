// k_csv_bytes.h -- k_csv_parse_bytes: the byte-parallel form of the device CSV tokenizer's common case (k_csv_pack.h holds the rules,
// the number / genre conversion and the exact line-at-a-time kernels this one falls back to).  Reference: get_dataset =
// make_csv_dataset(..., na_value="0", ignore_errors=True) (DeepFM.py:14-22) + the feature columns' id resolution (DeepFM.py:54-76).
// Included inside sparrow_hip.hip's anonymous namespace, after k_csv_pack.h.
//
// k_csv_parse<OPT> gives every LINE to a lane, which then walks its fields one after the other: 64 lanes in 64 different lines,
// dependent LDS reads, 61 % of the wave cycles in s_waitcnt (profiles/r02).  Here every lane takes 16 BYTES:
//   * one wave owns one 4-KB chunk of the text (the chunk k_csv_count counted: the scan of those counts is the line index at the
//     chunk's first byte) and is independent of every other wave -- no workgroup barrier after the tables are in LDS;
//   * the lane classifies its 16 bytes into an interleaved bit mask (bit 2i: byte i is ',', bit 2i+1: byte i is '\n') with exact
//     SWAR byte compares; (newlines, commas since the last newline) is an associative pair, scanned across the wave with DPP row
//     shifts / row broadcasts -- after that every lane knows the line index and the field index of each of its bytes;
//   * where the line in progress at the chunk's first byte began is found by classifying the 1 KB BEFORE the chunk the same way
//     (one extra pass of the wave; a line longer than that sends the call to the exact kernels);
//   * a lane then visits the field STARTS among its bytes (a byte after a ',' or a '\n'); the ones whose column a column list
//     names are converted by the same csv_emit as the line kernels -- the field's end comes from the lane's own mask when it lies
//     within its 16 bytes, from an 8-byte window search otherwise; fields of unnamed columns cost nothing at all;
//   * every '\n' checks its line's field count against the header's (ignore_errors=True drops other lines: any such line, like in
//     k_csv_parse<true>, makes the host rerun the exact keep -> scan -> parse sequence).
// Quotes.  The host splitter (split_csv_line, api_ingest.h) opens a quoted field only at a field start and lets it run to its
// closing quote; the byte-parallel form treats EVERY comma as a separator, which is the same thing exactly when no quoted field
// holds a comma -- so every '"' is checked locally: it must be either an opening quote (after ',' / '\n') whose next special
// character is a '"', or a closing quote (before ',' / '\n' / "\r\n") whose previous special character is a '"'.  That covers the
// reference's Spark-written files (an empty string is spelt "", 10 % of the rows); anything else (a comma or an escaped quote
// inside quotes, text after a closing quote, an unclosed quote) is counted in *drops: the exact kernels then decide.
#pragma once

#define PB_BYTES 4096                          // bytes per wave == CSV_CHUNK (k_csv_count's chunk)
#define PB_WAVES 4
#define PB_BACK 1024                           // how far before its chunk a wave looks for the start of the line in progress
#define PB_PRE 64                              // staged bytes before the chunk (backward quote check, '\r' before a '\n' at offset 0)
#define PB_POST 80                             // staged bytes after it (a field that starts inside the chunk may end there)
#define PB_PIECE (PB_PRE + PB_BYTES + PB_POST)
#define PB_FIELD_MAX 48                        // a named column's field longer than this goes to the exact kernels
static_assert(PB_BYTES == CSV_CHUNK, "one wave per counted chunk");
static_assert(PB_PIECE % 16 == 0, "16-byte staging");

// 0x80 in every byte of x that equals the byte replicated in pat (exact, no borrow between bytes)
__device__ __forceinline__ unsigned pb_eq(unsigned x, unsigned pat) {
    const unsigned t = x ^ pat;
    return ~((((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) | 0x7F7F7F7Fu);
}
// 16 bytes -> bit 2i: byte i is ',', bit 2i+1: byte i is '\n'; quotes = non-zero when one of them is '"'
__device__ __forceinline__ unsigned pb_masks(const uint4 w, unsigned& quotes) {
    const unsigned ww[4] = {w.x, w.y, w.z, w.w};
    unsigned M = 0;
    quotes = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned fn = pb_eq(ww[k], 0x0A0A0A0Au), fc = pb_eq(ww[k], 0x2C2C2C2Cu);
        quotes |= pb_eq(ww[k], 0x22222222u);
        unsigned y = (fn | (fc >> 1)) >> 6;            // byte b: comma at bit 8b, newline at bit 8b + 1
        y |= y >> 6;
        y |= y >> 12;
        M |= (y & 0xFFu) << (8 * k);
    }
    return M;
}
// a lane's 16 bytes as (newlines << 16 | commas after the last newline)
__device__ __forceinline__ unsigned pb_summary(unsigned M) {
    const unsigned nlm = M & 0xAAAAAAAAu, cm = M & 0x55555555u;
    const unsigned tail = nlm ? cm & ~((2u << (31 - __builtin_clz(nlm))) - 1u) : cm;
    return ((unsigned)__popc(nlm) << 16) | (unsigned)__popc(tail);
}
// (a then b): newlines add; b's commas count from its own last newline when it has one
__device__ __forceinline__ unsigned pb_comb(unsigned a, unsigned b) { return b + ((b >> 16) ? (a & 0xFFFF0000u) : a); }
// inclusive scan of pb_comb over the wave: row_shr 1 / 2 / 4 / 8, then lane 15 of a row into the next row, lane 31 into rows 2-3
__device__ __forceinline__ unsigned pb_scan(unsigned v) {
    v = pb_comb((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false), v);
    v = pb_comb((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false), v);
    v = pb_comb((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false), v);
    v = pb_comb((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false), v);
    v = pb_comb((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false), v);
    v = pb_comb((unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false), v);
    return v;
}
// the value of the lane below (lane 0: `first`)
__device__ __forceinline__ unsigned pb_from_below(unsigned v, unsigned first) {
    const unsigned o = (unsigned)__shfl_up((int)v, 1);
    return (threadIdx.x & 63) == 0 ? first : o;
}
// commas between the last newline below bit `bit` and that bit (none below: the lane's carry), newlines below it
__device__ __forceinline__ void pb_where(unsigned M, unsigned bit, unsigned st, unsigned& field, unsigned& lines_before) {
    const unsigned below = (1u << bit) - 1u;
    const unsigned nlb = M & 0xAAAAAAAAu & below, cb = M & 0x55555555u & below;
    lines_before = (st >> 16) + (unsigned)__popc(nlb);
    field = nlb ? (unsigned)__popc(cb & ~((2u << (31 - __builtin_clz(nlb))) - 1u)) : (st & 0xFFFFu) + (unsigned)__popc(cb);
}

__global__ __launch_bounds__(PB_WAVES * 64) void k_csv_parse_bytes(const CsvDev* __restrict__ Ld, const unsigned char* __restrict__ text, size_t len,
                                                                    int virt_nl, const unsigned* __restrict__ first, size_t n_first,
                                                                    const unsigned* __restrict__ total_nl, unsigned max_rows, int* __restrict__ ids,
                                                                    float* __restrict__ dense, unsigned long long* __restrict__ first_err,
                                                                    CsvErr* __restrict__ errs, unsigned* __restrict__ n_errs, unsigned* __restrict__ drops) {
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
    // LDS: the layout | the genre hash table | one staged piece per wave
    constexpr unsigned TAB_BYTES = (sizeof(CsvDev) + 15) & ~15u;
    const CsvDev& T = *reinterpret_cast<const CsvDev*>(lds);
    unsigned long long* g_tab = reinterpret_cast<unsigned long long*>(lds + TAB_BYTES);
    signed char* g_len = reinterpret_cast<signed char*>(g_tab + 64);
    for (unsigned i = threadIdx.x; i < sizeof(CsvDev) / 4; i += PB_WAVES * 64)
        reinterpret_cast<unsigned*>(lds)[i] = reinterpret_cast<const unsigned*>(Ld)[i];
    if (threadIdx.x < 32) {
        g_tab[2 * threadIdx.x] = Ld->gt_lo[threadIdx.x];
        g_tab[2 * threadIdx.x + 1] = Ld->gt_hi[threadIdx.x];
        g_len[threadIdx.x] = Ld->gt_len[threadIdx.x];
        g_len[32 + threadIdx.x] = Ld->gt_idx[threadIdx.x];
    }
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t chunk = (size_t)blockIdx.x * PB_WAVES + wave;
    const size_t base = chunk * PB_BYTES;
    const size_t len_eff = len + (virt_nl ? 1 : 0);               // (a text that does not end in '\n': a newline is imagined behind it)
    if (base >= len_eff) return;
    const unsigned L0 = chunk < n_first ? first[chunk] : *total_nl;   // index of the line in progress at `base` (line 0 = the header)
    if (L0 > max_rows) return;                                    // every field from here on belongs to a row >= max_rows
    unsigned char* piece = lds + TAB_BYTES + CSV_LDS_GENRE + wave * PB_PIECE;
    bool bad = false;

    // ---- the line in progress: commas since the last '\n' of the 1 KB before the chunk ----
    unsigned carry = 0, carry_edge = 3;                            // (file start: as if a '\n' came before)
    if (base > 0) {
        const uint4 w = *reinterpret_cast<const uint4*>(text + base - PB_BACK + 16 * lane);
        unsigned q;
        const unsigned M = pb_masks(w, q);
        const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)pb_scan(pb_summary(M)), 63);
        const unsigned D = (M | (M >> 1)) & 0x55555555u;
        carry_edge = (unsigned)__builtin_amdgcn_readlane((int)(((D >> 30) & 1u) | ((M >> 31) << 1)), 63);
        if ((tot >> 16) == 0) {                                   // no line start within PB_BACK bytes: the exact kernels take over
            if (lane == 0) atomicAdd(drops, 1u);
            return;
        }
        carry = tot & 0xFFFFu;
    }

    // ---- stage [base - PB_PRE, base + PB_BYTES + PB_POST) ----
    for (unsigned i = lane; i < PB_PIECE / 16; i += 64) {
        const long long g = (long long)base - PB_PRE + 16ll * i;
        uint4 w;
        if (g < 0) {
            w = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
        } else if ((size_t)g + 16 <= len) {
            w = *reinterpret_cast<const uint4*>(text + g);
        } else {
            unsigned ww[4] = {0, 0, 0, 0};
            for (int b = 0; b < 16; ++b) {
                const size_t p = (size_t)g + b;
                const unsigned c = p < len ? text[p] : (p == len && virt_nl ? '\n' : 0);
                ww[b >> 2] |= c << (8 * (b & 3));
            }
            w = make_uint4(ww[0], ww[1], ww[2], ww[3]);
        }
        *reinterpret_cast<uint4*>(piece + 16 * i) = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const CsvRdLds rd{piece};
    const CsvSink sink{ids, dense, first_err, errs, n_errs};
    const int n_cols = T.n_cols;
#pragma unroll 1
    for (unsigned pass = 0; pass < PB_BYTES / 1024; ++pass) {
        const unsigned o = pass * 1024 + 16 * lane;               // offset in the chunk
        const size_t gpos = base + o;
        if (base + pass * 1024 >= len_eff) break;                 // (wave-uniform)
        const uint4 w = *reinterpret_cast<const uint4*>(piece + PB_PRE + o);
        unsigned qflags;
        const unsigned M = pb_masks(w, qflags);
        const unsigned incl = pb_scan(pb_summary(M));
        const unsigned st = pb_comb(carry, pb_from_below(incl, 0u));        // state in front of this lane's first byte
        const unsigned D = (M | (M >> 1)) & 0x55555555u;
        const unsigned edge = ((D >> 30) & 1u) | ((M >> 31) << 1);
        const unsigned prev = pb_from_below(edge, carry_edge);
        carry = pb_comb(carry, (unsigned)__builtin_amdgcn_readlane((int)incl, 63));
        carry_edge = (unsigned)__builtin_amdgcn_readlane((int)edge, 63);

        // ---- every '\n': does its line have the header's field count? ----
        for (unsigned m = (M >> 1) & 0x55555555u; m; m &= m - 1) {
            const unsigned bit = (unsigned)__builtin_ctz(m);
            unsigned k, lb;
            pb_where(M, bit, st, k, lb);
            if (L0 + lb > max_rows) continue;                     // (lines behind the last row asked for are not examined)
            if (n_cols == 1) {                                    // one column: only an empty line is not a row
                const unsigned p = PB_PRE + o + (bit >> 1);
                const unsigned c1 = piece[p - 1], c2 = piece[p - 2];
                bad |= c1 == '\n' || (c1 == '\r' && c2 == '\n');
            } else {
                bad |= (int)k != n_cols - 1;
            }
        }

        // ---- every '"' is an opening or a closing quote of a field without separators inside ----
        if (qflags) {
            for (unsigned i = 0; i < 16; ++i) {
                const unsigned p = PB_PRE + o + i;
                if (piece[p] != '"' || gpos + i >= len) continue;
                const unsigned pb = piece[p - 1], nb = piece[p + 1], nb2 = piece[p + 2];
                const bool is_open = pb == ',' || pb == '\n';
                const bool is_close = nb == ',' || nb == '\n' || (nb == '\r' && nb2 == '\n');
                if (is_open == is_close) { bad = true; continue; }
                bool ok = false;
                for (unsigned d = 1; d <= PB_FIELD_MAX; ++d) {
                    if (!is_open && p < d) break;
                    const unsigned c = is_open ? piece[p + d] : piece[p - d];
                    if (c == '"') { ok = true; break; }
                    if (c == ',' || c == '\n') break;
                }
                bad |= !ok;
            }
        }

        // ---- field starts ----
        for (unsigned m = ((D << 2) | (prev & 1u)) & 0x55555555u; m; m &= m - 1) {
            const unsigned bit = (unsigned)__builtin_ctz(m);
            const unsigned j = bit >> 1;
            if (gpos + j >= len_eff) break;
            unsigned k, lb;
            pb_where(M, bit, st, k, lb);
            const unsigned line = L0 + lb;
            if (line == 0 || line - 1 >= max_rows || (int)k >= n_cols) continue;
            const int role = T.role[k];
            if (!role) continue;
            unsigned a = PB_PRE + o + j, b;
            if (piece[a] == '"') {                                // quoted: the content runs to the closing quote
                ++a;
                b = a;
                unsigned tries = 0;
                for (; tries < PB_FIELD_MAX / 8; ++tries) {
                    const unsigned f = csv_find8(rd.win(b), '"');
                    b += f;
                    if (f < 8) break;
                }
                if (tries == PB_FIELD_MAX / 8) { bad = true; continue; }
            } else {
                const unsigned Dh = D >> bit;
                bool nl_end;
                if (Dh) {                                         // the separator is among this lane's bytes
                    const unsigned d2 = (unsigned)__builtin_ctz(Dh);
                    b = a + (d2 >> 1);
                    nl_end = (M >> (bit + d2 + 1)) & 1u;
                } else {
                    b = a + (16 - j);
                    unsigned tries = 0;
                    for (; tries < PB_FIELD_MAX / 8; ++tries) {
                        const unsigned long long w8 = rd.win(b);
                        const unsigned f1 = csv_find8(w8, ','), f2 = csv_find8(w8, '\n');
                        const unsigned f = f1 < f2 ? f1 : f2;
                        b += f;
                        if (f < 8) break;
                    }
                    if (tries == PB_FIELD_MAX / 8) { bad = true; continue; }
                    nl_end = piece[b] == '\n';
                }
                if (nl_end && b > a && piece[b - 1] == '\r') --b;
            }
            csv_emit(T, g_tab, g_len, rd, a, b, false, role, (int)k, line - 1, sink);
        }
    }
    if (bad) atomicAdd(drops, 1u);
}

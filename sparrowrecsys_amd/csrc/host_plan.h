// host_plan.h -- plan validation and small host helpers.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
namespace {

int lds_stride(int width) {
    // floats per sample row in LDS: multiple of 4 with (stride/4) odd, so the 16-B slots of the 16
    // rows a ds_read_b128 lane group touches spread over the 256-B bank row
    int s = (width + 3) & ~3;
    if (((s >> 2) & 1) == 0) s += 4;
    return s;
}

int check_slot(const sprk_plan& p, int slot, bool allow_none, const char* what) {
    if (slot == -1 && allow_none) return 0;
    if (slot < 0 || slot >= p.n_slots) return fail(SPRK_EINVAL, "%s: slot %d outside [0,%d)", what, slot, p.n_slots);
    return 0;
}

int validate_plan(const sprk_plan& p) {
    if (p.abi_version != SPRK_ABI_VERSION) return fail(SPRK_EINVAL, "plan abi_version %d != %d", p.abi_version, SPRK_ABI_VERSION);
    if (p.n_id_cols < 0 || p.n_dense < 0 || p.n_aux < 0) return fail(SPRK_EINVAL, "negative column count");
    if (p.n_bufs < 1 || p.n_bufs > SPRK_MAX_BUFS) return fail(SPRK_EINVAL, "n_bufs %d outside [1,%d]", p.n_bufs, SPRK_MAX_BUFS);
    if (p.n_segs < 0 || p.n_segs > SPRK_MAX_SEGS) return fail(SPRK_EINVAL, "n_segs %d too large", p.n_segs);
    if (p.n_ops < 0 || p.n_ops > SPRK_MAX_OPS) return fail(SPRK_EINVAL, "n_ops %d too large", p.n_ops);
    if (p.n_taps < 0 || p.n_taps > SPRK_MAX_TAPS) return fail(SPRK_EINVAL, "n_taps %d too large", p.n_taps);
    if (p.n_pairs < 0 || p.n_pairs > SPRK_MAX_PAIRS) return fail(SPRK_EINVAL, "n_pairs %d too large", p.n_pairs);
    if (p.n_slots < 0 || p.n_slots > 4096) return fail(SPRK_EINVAL, "n_slots %d out of range", p.n_slots);
    for (int b = 0; b < p.n_bufs; ++b)
        if (p.buf_width[b] <= 0 || (p.buf_width[b] & 3)) return fail(SPRK_EINVAL, "buf_width[%d]=%d must be a positive multiple of 4", b, p.buf_width[b]);
    for (int i = 0; i < p.n_segs; ++i) {
        const sprk_seg& s = p.segs[i];
        int width = 0;
        switch (s.kind) {
            case SPRK_SEG_ROWS:
            case SPRK_SEG_CROSS_ROWS:
                if (check_slot(p, s.slot, false, "segment table")) return SPRK_EINVAL;
                if (s.row_stride <= 0 || (s.row_stride & 3) || s.count <= 0 || 4 * s.count > s.row_stride || (s.dst & 3))
                    return fail(SPRK_EINVAL, "segment %d: bad row geometry (row_stride %d, count %d, dst %d)", i, s.row_stride, s.count, s.dst);
                width = 4 * s.count;
                break;
            case SPRK_SEG_SCALAR:
            case SPRK_SEG_CROSS_SCALAR:
                if (check_slot(p, s.slot, false, "segment table")) return SPRK_EINVAL;
                width = 1;
                break;
            case SPRK_SEG_DENSE:
                if (s.count <= 0 || s.field < 0 || s.field + s.count > p.n_dense) return fail(SPRK_EINVAL, "segment %d: dense columns out of range", i);
                width = s.count;
                break;
            case SPRK_SEG_AUX:
                if (s.count <= 0 || s.field < 0 || s.field + s.count > p.n_aux) return fail(SPRK_EINVAL, "segment %d: aux columns out of range", i);
                width = s.count;
                break;
            case SPRK_SEG_ZERO:
                if (s.count <= 0) return fail(SPRK_EINVAL, "segment %d: empty zero fill", i);
                width = s.count;
                break;
            default:
                return fail(SPRK_EINVAL, "segment %d: unknown kind %d", i, s.kind);
        }
        if (s.kind == SPRK_SEG_ROWS || s.kind == SPRK_SEG_SCALAR) {
            if (s.field < 0 || s.field >= p.n_id_cols) return fail(SPRK_EINVAL, "segment %d: ids column %d out of range", i, s.field);
            if (s.vocab <= 0) return fail(SPRK_EINVAL, "segment %d: vocab must be positive", i);
        }
        if (s.kind == SPRK_SEG_CROSS_ROWS || s.kind == SPRK_SEG_CROSS_SCALAR) {
            if (s.field < 0 || s.field >= p.n_id_cols || s.field2 < 0 || s.field2 >= p.n_id_cols)
                return fail(SPRK_EINVAL, "segment %d: cross ids columns out of range", i);
            if (s.vocab <= 0) return fail(SPRK_EINVAL, "segment %d: bucket count must be positive", i);
        }
        if (s.dst < 0 || s.dst + width > p.buf_width[0]) return fail(SPRK_EINVAL, "segment %d: writes [%d,%d) outside buffer 0 (width %d)", i, s.dst, s.dst + width, p.buf_width[0]);
    }
    for (int i = 0; i < p.n_ops; ++i) {
        const sprk_op& o = p.ops[i];
        if (o.src_buf < 0 || o.src_buf >= p.n_bufs || o.dst_buf < 0 || o.dst_buf >= p.n_bufs) return fail(SPRK_EINVAL, "op %d: buffer index out of range", i);
        if (o.kind == SPRK_OP_DENSE) {
            if (o.src_buf == o.dst_buf) return fail(SPRK_EINVAL, "op %d: Dense must not run in place", i);
            if (o.K <= 0 || (o.K & 3) || o.N <= 0 || (o.N & 15) || o.ldw < o.K || (o.ldw & 3)) return fail(SPRK_EINVAL, "op %d: bad Dense geometry K=%d N=%d ldw=%d", i, o.K, o.N, o.ldw);
            if ((o.src_off & 3) || (o.dst_off & 3)) return fail(SPRK_EINVAL, "op %d: offsets must be multiples of 4", i);
            if (o.src_off < 0 || o.src_off + o.K > p.buf_width[o.src_buf] || o.dst_off < 0 || o.dst_off + o.N > p.buf_width[o.dst_buf]) return fail(SPRK_EINVAL, "op %d: Dense slice outside its buffer", i);
            if (check_slot(p, o.w_slot, false, "Dense kernel") || check_slot(p, o.b_slot, false, "Dense bias")) return SPRK_EINVAL;
            if (o.act == SPRK_ACT_PRELU && check_slot(p, o.alpha_slot, false, "PReLU alpha")) return SPRK_EINVAL;
            if (o.act < 0 || o.act > SPRK_ACT_PRELU) return fail(SPRK_EINVAL, "op %d: unknown activation", i);
        } else if (o.kind == SPRK_OP_FM_SUMSQ) {
            if (o.K <= 0 || o.groups <= 0 || o.group_stride < o.K) return fail(SPRK_EINVAL, "op %d: bad FM geometry", i);
            if (o.src_off < 0 || o.src_off + (o.groups - 1) * o.group_stride + o.K > p.buf_width[o.src_buf] || o.dst_off < 0 || o.dst_off + o.K > p.buf_width[o.dst_buf]) return fail(SPRK_EINVAL, "op %d: FM slice outside its buffer", i);
            if (o.src_buf == o.dst_buf && o.dst_off < o.src_off + (o.groups - 1) * o.group_stride + o.K && o.dst_off + o.K > o.src_off) return fail(SPRK_EINVAL, "op %d: FM output overlaps its input", i);
        } else if (o.kind == SPRK_OP_PAIR_DOT) {
            if (o.K <= 0 || (o.K & 3) || p.n_pairs <= 0) return fail(SPRK_EINVAL, "op %d: bad pair-dot geometry", i);
            for (int j = 0; j < p.n_pairs; ++j)
                if (p.pair_a[j] < 0 || (p.pair_a[j] & 3) || p.pair_a[j] + o.K > p.buf_width[o.src_buf] || p.pair_b[j] < 0 || (p.pair_b[j] & 3) || p.pair_b[j] + o.K > p.buf_width[o.src_buf]) return fail(SPRK_EINVAL, "op %d: pair %d outside its buffer", i, j);
            if (o.dst_off < 0 || o.dst_off + p.n_pairs > p.buf_width[o.dst_buf]) return fail(SPRK_EINVAL, "op %d: pair-dot output outside its buffer", i);
        } else {
            return fail(SPRK_EINVAL, "op %d: unknown kind %d", i, o.kind);
        }
    }
    for (int i = 0; i < p.n_taps; ++i) {
        const sprk_tap& t = p.taps[i];
        if (t.buf < 0 || t.buf >= p.n_bufs || t.off < 0 || t.len <= 0 || t.off + t.len > p.buf_width[t.buf]) return fail(SPRK_EINVAL, "tap %d outside its buffer", i);
        if (check_slot(p, t.w_slot, true, "tap weights")) return SPRK_EINVAL;
    }
    if (p.din.enabled == 2) {
        const sprk_din& d = p.din;
        if (d.T <= 0 || d.T > 256) return fail(SPRK_EINVAL, "DIEN history length %d outside [1,256]", d.T);
        if (d.hist_col < 0 || d.hist_col + d.T > p.n_id_cols || d.cand_col < 0 || d.cand_col >= p.n_id_cols) return fail(SPRK_EINVAL, "DIEN ids columns out of range");
        if (d.row_stride <= 0 || (d.row_stride & 3) || d.vocab <= 0) return fail(SPRK_EINVAL, "DIEN bad table geometry");
        if ((d.emb_dim != 10 && d.emb_dim != 16) || d.emb_dim > d.row_stride) return fail(SPRK_EINVAL, "DIEN emb_dim %d: instantiated for 10 and 16", d.emb_dim);
        if (d.hidden != 32) return fail(SPRK_EINVAL, "DIEN attention width must be 32 (DIEN.py:184)");
        if (p.n_aux != d.row_stride) return fail(SPRK_EINVAL, "DIEN: n_aux (%d) must equal row_stride (%d)", p.n_aux, d.row_stride);
        if (check_slot(p, d.table_slot, false, "DIEN table") || check_slot(p, d.seq_slot, false, "DIEN sequence weights")) return SPRK_EINVAL;
    } else if (p.din.enabled) {
        const sprk_din& d = p.din;
        if (p.din.enabled != 1) return fail(SPRK_EINVAL, "din.enabled must be 0, 1 (DIN) or 2 (DIEN)");
        if (d.T <= 0 || d.T > 256) return fail(SPRK_EINVAL, "DIN history length %d outside [1,256]", d.T);
        if (d.hist_col < 0 || d.hist_col + d.T > p.n_id_cols || d.cand_col < 0 || d.cand_col >= p.n_id_cols) return fail(SPRK_EINVAL, "DIN ids columns out of range");
        if (d.row_stride <= 0 || (d.row_stride & 3) || d.vocab <= 0) return fail(SPRK_EINVAL, "DIN bad table geometry");
        if (d.hidden <= 0 || (d.hidden & 15)) return fail(SPRK_EINVAL, "DIN hidden width must be a multiple of 16");
        if (p.n_aux != d.row_stride) return fail(SPRK_EINVAL, "DIN: n_aux (%d) must equal row_stride (%d)", p.n_aux, d.row_stride);
        if (check_slot(p, d.table_slot, false, "DIN table") || check_slot(p, d.w_slot, false, "DIN att0 kernel") || check_slot(p, d.b_slot, false, "DIN att0 bias") || check_slot(p, d.alpha_slot, false, "DIN alpha") || check_slot(p, d.w2_slot, false, "DIN att1 kernel")) return SPRK_EINVAL;
    } else if (p.n_aux != 0) {
        return fail(SPRK_EINVAL, "n_aux %d without a DIN stage", p.n_aux);
    }
    return SPRK_OK;
}



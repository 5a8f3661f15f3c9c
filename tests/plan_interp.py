"""TEST INFRASTRUCTURE: a numpy interpreter of ``sprk_plan`` + weight slots.

Executes the same segment / op / tap lists the HIP kernels execute, in float64 or float32, so the
host-side plan compiler (layout permutations, padding, first-order offsets, tap scales) can be
checked against the oracle on a machine without a GPU.  It is NOT a product path and lives under
tests/ on purpose; nothing in ``sparrowrecsys_amd`` imports it.
"""
import numpy as np

from sparrowrecsys_amd import _lib as L

_U64 = (1 << 64) - 1


def _cross_bucket(a, b, buckets):
    k_mul = np.uint64(0xC6A4A7935BD1E995)
    s47 = np.uint64(47)
    with np.errstate(over="ignore"):
        h = np.full(len(a), 0xDECAFCAFFE, dtype=np.uint64)
        for col in (a, b):
            v = np.asarray(col).astype(np.int64).astype(np.uint64)
            r = h ^ k_mul
            t = v * k_mul
            t = (t ^ (t >> s47)) * k_mul
            r = r ^ t
            r = r * k_mul
            r = (r ^ (r >> s47)) * k_mul
            r = r ^ (r >> s47)
            h = r
        return (h % np.uint64(buckets)).astype(np.int64)


def _slot(slots, i):
    a = slots[i]
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def din_pool(plan, slots, ids, dtype=np.float64):
    d = plan.din
    T, Dp, Hp = d.T, d.row_stride, d.hidden
    table = _slot(slots, d.table_slot).reshape(-1, Dp).astype(dtype)
    W = _slot(slots, d.w_slot).reshape(Hp, 4 * Dp).astype(dtype)
    b = _slot(slots, d.b_slot).astype(dtype)
    alpha = _slot(slots, d.alpha_slot).reshape(T, Hp).astype(dtype)
    w2 = _slot(slots, d.w2_slot).astype(dtype)
    hist = ids[:, d.hist_col:d.hist_col + T]
    cand = ids[:, d.cand_col]
    h = table[hist]
    c = np.repeat(table[cand][:, None, :], T, axis=1)
    a = np.concatenate([h - c, h, c, h * c], axis=-1)
    u = a @ W.T + b
    u = np.maximum(u, 0) + alpha[None] * np.minimum(u, 0)
    s = u @ w2 + dtype(d.b2)
    att = 1.0 / (1.0 + np.exp(-s))
    pooled = (h * att[..., None]).sum(axis=1)
    return pooled, att


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def dien_seq(plan, slots, ids, dtype=np.float64):
    """DIEN stage (din.enabled == 2) from the PACKED weight image, layout as documented in include/sparrow_hip.h
    (sprk_din.seq_slot) -- parsed here independently of models.DIEN._seq_image."""
    d = plan.din
    T, Dp, D, H = d.T, d.row_stride, d.emb_dim, d.hidden
    Dq, N3 = (D + 3) // 4 * 4, (3 * D + 3) // 4 * 4
    img = _slot(slots, d.seq_slot).astype(dtype)
    pos = [0]

    def take(rows, stride, cols):
        a = img[pos[0]:pos[0] + rows * stride].reshape(rows, stride)[:, :cols]
        pos[0] += rows * stride
        return a
    Wk, Uk, bk = take(D, N3, 3 * D), take(D, N3, 3 * D), take(2, N3, 3 * D)
    A0, a0b, a1, a1b = take(D, H, H), take(1, H, H)[0], take(1, H, H)[0], take(1, 4, 1)[0, 0]
    gates = []
    for _ in range(3):
        gates.append((take(D, Dq, D), take(1, Dq, D)[0], take(D, Dq, D), take(D, Dq, D), take(1, Dq, D)[0]))
    h0 = take(1, Dq, D)[0]
    assert (-pos[0]) % 64 + pos[0] == img.size
    table = _slot(slots, d.table_slot).reshape(-1, Dp).astype(dtype)[:, :D]
    hist = ids[:, d.hist_col:d.hist_col + T]
    c = table[ids[:, d.cand_col]]
    B = ids.shape[0]
    h = np.zeros((B, D), dtype)
    g = np.zeros((B, D), dtype)
    hs = np.repeat(h0[None, :], B, axis=0)

    def gate(k, gt, hid, act):
        ik, ib, hk, ok, ob = gates[k]
        return act((gt @ ik + ib + hid @ hk) @ ok + ob)
    for t in range(T):
        x = table[hist[:, t]]
        mx, mh = x @ Wk + bk[0], h @ Uk + bk[1]
        z, r = _sig(mx[:, :D] + mh[:, :D]), _sig(mx[:, D:2 * D] + mh[:, D:2 * D])
        hn = z * h + (1 - z) * np.tanh(mx[:, 2 * D:] + r * mh[:, 2 * D:])
        live = (hist[:, t] != 0)[:, None]
        h = np.where(live, hn, h)
        g = np.where(live, hn, g)
        a = _sig(_sig((g * c) @ A0 + a0b) @ a1 + a1b)[:, None]
        r_t, z_t = gate(0, g, hs, _sig), gate(1, g, hs, _sig)
        hnext = gate(2, g, hs * z_t, np.tanh)
        u = a * r_t
        hs = (1 - u) * hs + u * hnext
    out = np.zeros((B, Dp), dtype)
    out[:, :D] = hs
    return out


def run_plan(plan, slots, ids, dense, dtype=np.float64):
    """-> scores [B] (float64/32).  ids [B,F] int, dense [B,N] float."""
    B = ids.shape[0] if plan.n_id_cols else dense.shape[0]
    aux = None
    if plan.din.enabled == 2:
        aux = dien_seq(plan, slots, ids, dtype)
    elif plan.din.enabled:
        aux, _ = din_pool(plan, slots, ids, dtype)
    bufs = [np.full((B, plan.buf_width[i]), np.nan, dtype=dtype) for i in range(plan.n_bufs)]
    for i in range(plan.n_segs):
        s = plan.segs[i]
        if s.kind in (L.SEG_ROWS, L.SEG_CROSS_ROWS):
            table = _slot(slots, s.slot).reshape(-1, s.row_stride).astype(dtype)
            if s.kind == L.SEG_ROWS:
                idx = ids[:, s.field].astype(np.int64)
                assert ((idx >= -1) & (idx < s.vocab)).all(), "id out of range in segment %d" % i
            else:
                idx = _cross_bucket(ids[:, s.field], ids[:, s.field2], s.vocab)
            rows = np.zeros((B, 4 * s.count), dtype=dtype)
            ok = idx >= 0
            rows[ok] = table[idx[ok], :4 * s.count]
            bufs[0][:, s.dst:s.dst + 4 * s.count] = rows
        elif s.kind in (L.SEG_SCALAR, L.SEG_CROSS_SCALAR):
            table = _slot(slots, s.slot).reshape(-1).astype(dtype)
            if s.kind == L.SEG_SCALAR:
                assert table.shape[0] == s.vocab + 1 and table[s.vocab] == 0
                idx = ids[:, s.field].astype(np.int64)
                assert ((idx >= -1) & (idx < s.vocab)).all()
            else:
                idx = _cross_bucket(ids[:, s.field], ids[:, s.field2], s.vocab)
            v = np.zeros(B, dtype=dtype)
            ok = idx >= 0
            v[ok] = table[idx[ok]]
            bufs[0][:, s.dst] = v
        elif s.kind == L.SEG_DENSE:
            bufs[0][:, s.dst:s.dst + s.count] = dense[:, s.field:s.field + s.count].astype(dtype)
        elif s.kind == L.SEG_AUX:
            bufs[0][:, s.dst:s.dst + s.count] = aux[:, s.field:s.field + s.count]
        elif s.kind == L.SEG_ZERO:
            bufs[0][:, s.dst:s.dst + s.count] = 0
        else:
            raise AssertionError("bad segment kind")
    for i in range(plan.n_ops):
        o = plan.ops[i]
        src, dst = bufs[o.src_buf], bufs[o.dst_buf]
        if o.kind == L.OP_DENSE:
            Wt = _slot(slots, o.w_slot).reshape(o.N, o.ldw).astype(dtype)[:, :o.K]
            b = _slot(slots, o.b_slot).astype(dtype)[:o.N]
            x = src[:, o.src_off:o.src_off + o.K]
            assert not np.isnan(x).any(), "op %d reads uninitialised LDS" % i
            y = x @ Wt.T + b
            if o.act == L.ACT_RELU:
                y = np.maximum(y, 0)
            elif o.act == L.ACT_PRELU:
                al = _slot(slots, o.alpha_slot).astype(dtype)[:o.N]
                y = np.maximum(y, 0) + al * np.minimum(y, 0)
            dst[:, o.dst_off:o.dst_off + o.N] = y
        elif o.kind == L.OP_FM_SUMSQ:
            v = np.stack([src[:, o.src_off + g * o.group_stride:o.src_off + g * o.group_stride + o.K]
                          for g in range(o.groups)], axis=1)
            assert not np.isnan(v).any()
            s = v.sum(axis=1)
            dst[:, o.dst_off:o.dst_off + o.K] = s * s - (v * v).sum(axis=1)
        elif o.kind == L.OP_PAIR_DOT:
            for p in range(plan.n_pairs):
                a = src[:, plan.pair_a[p]:plan.pair_a[p] + o.K]
                b = src[:, plan.pair_b[p]:plan.pair_b[p] + o.K]
                assert not (np.isnan(a).any() or np.isnan(b).any())
                dst[:, o.dst_off + p] = (a * b).sum(axis=1)
        else:
            raise AssertionError("bad op kind")
    z = np.full(B, plan.head_bias, dtype=dtype)
    for i in range(plan.n_taps):
        t = plan.taps[i]
        x = bufs[t.buf][:, t.off:t.off + t.len]
        assert not np.isnan(x).any(), "tap %d reads uninitialised LDS" % i
        if t.w_slot >= 0:
            s = x @ _slot(slots, t.w_slot).astype(dtype)[:t.len]
        else:
            s = x.sum(axis=1)
        z = z + dtype(t.scale) * (s + dtype(t.bias))
    return 1.0 / (1.0 + np.exp(-z))

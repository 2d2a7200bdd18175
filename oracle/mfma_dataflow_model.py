"""TEST INFRASTRUCTURE ONLY: a lane-level numpy model of the register dataflow of csrc/mlp_head.hip (the fused feature -> 3-layer
tanh MLP -> 3x4 affine head of the neural bilateral variants, /root/reference/project/models/modules.py:595-820 and the trainer's
application models/trainers/scene_graph.py:99-106).

It executes, for ONE wave and ONE tile of 32 pixels, exactly the sequence of v_mfma_f32_32x32x2_f32 operations the kernel issues,
with every register modelled as an array of 64 lanes and the instruction modelled by its documented operand layout
(/opt/skills/guides/cdna_hip_programming.md section 3):

    A operand: lane l holds A[i = l & 31][k = l >> 5]        B operand: lane l holds B[k = l >> 5][j = l & 31]
    C / D    : lane l, register r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]

so that the index maps of the kernel (which weight goes to which lane in which step, where the 12 affine entries land, how the
tiles are transposed through LDS for the weight gradients) are checked on the CPU against a plain matrix implementation
(tests/test_mlp_head_dataflow.py) before the kernel ever runs.  The chaining trick it proves: with activations kept TRANSPOSED
(neurons x pixels), the D registers of one layer ARE the B operands of the next -- step s of the next layer consumes register s
of the tile, i.e. neuron rowmap(s, half) -- because the order of the k summation is free as long as the A operand (the weights,
read from LDS in any order we like) follows the same map.  No cross-lane movement between the layers."""
from __future__ import annotations

import numpy as np

f32 = np.float32
LANES = np.arange(64)
COL = LANES & 31          # pixel of the tile a lane works for
HALF = LANES >> 5
HID = 64                  # hidden width (2 row blocks of 32)
OUTP = 12                 # affine entries (padded to one 32-row block)


def rowmap(r, half):
    """Row of a D tile held in register r by a lane of the given half."""
    return (r & 3) + 8 * (r >> 2) + 4 * half


def mfma_32x32x2(a, b, c):
    """a, b: [64] lanes; c: [16, 64] (register, lane).  Returns d = A B + C in the same layout."""
    A = np.zeros((32, 2), f32); B = np.zeros((2, 32), f32)
    A[COL, HALF] = a
    B[HALF, COL] = b
    D = (A @ B).astype(f32)
    d = c.copy()
    for r in range(16):
        d[r] += D[rowmap(r, HALF), COL]
    return d


def zeros_tile():
    return np.zeros((16, 64), f32)


def tile_to_matrix(t, rows=32):
    """[16, 64] D-layout registers -> [rows, 32 px] matrix."""
    M = np.zeros((32, 32), f32)
    for r in range(16):
        M[rowmap(r, HALF), COL] = t[r]
    return M[:rows]


# ---------------------------------------------------------------------------------------------------------------------------------
# forward
# ---------------------------------------------------------------------------------------------------------------------------------
def load_features(feats):
    """feats [32 px, F] -> KS1 = F/2 B-operand registers: lane (px, h) holds feature k1(s, h) = s + KS1 * h in step s (each lane reads
    KS1 consecutive floats of its pixel's row)."""
    F = feats.shape[1]
    KS1 = F // 2
    return [feats[COL, s + KS1 * HALF].astype(f32) for s in range(KS1)]


def layer_first(W1, x):
    """W1 [64, F] (torch Linear layout [out, in]); x: KS1 registers.  Returns 2 D tiles (row blocks of the 64 neurons)."""
    KS1 = len(x)
    out = []
    for o in range(2):
        acc = zeros_tile()
        for s in range(KS1):
            a = W1[32 * o + COL, s + KS1 * HALF]                 # A[i = out neuron][k = k1(s, half)]
            acc = mfma_32x32x2(a.astype(f32), x[s], acc)
        out.append(acc)
    return out


def layer_chained(W, tiles, n_out_blocks):
    """W [out, in]; tiles: the previous layer's D tiles (in = 32 b + rowmap(s, half)).  The D registers are the B operands."""
    out = []
    for o in range(n_out_blocks):
        acc = zeros_tile()
        for b, t in enumerate(tiles):
            for s in range(16):
                rows = 32 * o + COL
                a = np.where(rows < W.shape[0], W[np.minimum(rows, W.shape[0] - 1), 32 * b + rowmap(s, HALF)], 0).astype(f32)
                acc = mfma_32x32x2(a, t[s], acc)
        out.append(acc)
    return out


def apply_affine(aff_tile, rgb, residual=True):
    """aff_tile: D tile of the 12 entries (row e = 4 r + c).  Half 0 lanes hold rows 0-3 (regs 0-3) and 8-11 (regs 4-7), half 1
    lanes rows 4-7 (regs 0-3): output channel 0 and 2 are produced by the half-0 lane of a pixel, channel 1 by its half-1 lane.
    Returns out [32 px, 3]."""
    out = np.zeros((32, 3), f32)
    px_rgb = rgb[COL]                                            # every lane has its pixel's colour
    def row(regs):
        return (aff_tile[regs[0]] * px_rgb[:, 0] + aff_tile[regs[1]] * px_rgb[:, 1] + aff_tile[regs[2]] * px_rgb[:, 2] + aff_tile[regs[3]]).astype(f32)
    lo = row((0, 1, 2, 3))          # half 0: channel 0, half 1: channel 1
    hi = row((4, 5, 6, 7))          # half 0: channel 2
    h0, h1 = HALF == 0, HALF == 1
    out[COL[h0], 0] = lo[h0]; out[COL[h1], 1] = lo[h1]; out[COL[h0], 2] = hi[h0]
    if residual:
        out += rgb
    return out


def forward(feats, rgb, W1, W2, W3, residual=True):
    x = load_features(feats)
    h1 = [np.tanh(t).astype(f32) for t in layer_first(W1, x)]
    h2 = [np.tanh(t).astype(f32) for t in layer_chained(W2, h1, 2)]
    aff = layer_chained(W3, h2, 1)[0]
    return apply_affine(aff, rgb, residual), dict(x=x, h1=h1, h2=h2, aff=aff)


# ---------------------------------------------------------------------------------------------------------------------------------
# backward
# ---------------------------------------------------------------------------------------------------------------------------------
def affine_grad_tile(v_out, rgb):
    """d(aff) in the D layout of the affine tile: entry 4 r + c gets v_out[r] * (rgb[c] | 1).  Rows 12..31 are zero."""
    t = zeros_tile()
    g = v_out[COL]; c = rgb[COL]
    h0 = (HALF == 0)
    r_lo = np.where(h0, g[:, 0], g[:, 1])                        # regs 0-3: row r = 0 (half 0) / r = 1 (half 1)
    for k in range(3):
        t[k] = r_lo * c[:, k]
    t[3] = r_lo
    r_hi = np.where(h0, g[:, 2], 0)                              # regs 4-7: r = 2 (half 0); half 1 would be entries 12-15
    for k in range(3):
        t[4 + k] = r_hi * c[:, k]
    t[7] = r_hi
    return t


def rgb_grad(aff_tile, v_out, residual=True):
    """v_rgb[c] = sum_r aff[r][c] v_out[r] (+ v_out[c]): each half contributes the rows it holds; the two partial sums of a pixel
    are added across the halves (one lane exchange with lane ^ 32)."""
    g = v_out[COL]
    h0 = (HALF == 0)
    r_lo = np.where(h0, g[:, 0], g[:, 1]); r_hi = np.where(h0, g[:, 2], 0)
    part = np.stack([aff_tile[k] * r_lo + aff_tile[4 + k] * r_hi for k in range(3)], axis=1).astype(f32)   # [64, 3]
    tot = part + part[LANES ^ 32]
    out = tot[:32].copy()
    if residual:
        out += v_out
    return out


def chained_transposed(W, tiles, n_out_blocks, k_regs=16):
    """d(in)^T = W^T d(out)^T: A[i = in neuron][k = out row], B = the D tiles of d(out).  k_regs < 16 when the upper registers of
    the tile are known zeros (the affine tile: 8)."""
    out = []
    for o in range(n_out_blocks):
        acc = zeros_tile()
        for b, t in enumerate(tiles):
            for s in range(k_regs):
                rows = 32 * b + rowmap(s, HALF)                  # out index of W
                cols = 32 * o + COL                              # in index of W
                ok = (rows < W.shape[0]) & (cols < W.shape[1])
                a = np.where(ok, W[np.minimum(rows, W.shape[0] - 1), np.minimum(cols, W.shape[1] - 1)], 0).astype(f32)
                acc = mfma_32x32x2(a, t[s], acc)
        out.append(acc)
    return out


LDS_STRIDE = 36


def tile_to_lds(tiles):
    """D tiles -> LDS image T[row][px] (row stride 36 floats): lane (px, h) writes register r of block b to row 32 b + rowmap(r, h)."""
    T = np.full((32 * len(tiles), LDS_STRIDE), np.nan, f32)
    for b, t in enumerate(tiles):
        for r in range(16):
            T[32 * b + rowmap(r, HALF), COL] = t[r]
    return T


def features_to_lds(x):
    """The first layer's B operands -> T[feature][px] (rows >= F stay unwritten: they only feed unused gradient columns)."""
    KS1 = len(x)
    T = np.full((32, LDS_STRIDE), np.nan, f32)
    for s in range(KS1):
        T[s + KS1 * HALF, COL] = x[s]
    return T


def outer_over_pixels(TU, TV, acc, n_u_blocks, n_v_blocks):
    """acc[ob][vb] += U V^T over the 32 pixels of the tile: step s takes pixel 16 kk + s from the lane half kk, on both operands
    (each lane reads 16 consecutive floats of its row)."""
    for ob in range(n_u_blocks):
        for vb in range(n_v_blocks):
            for s in range(16):
                a = TU[32 * ob + COL, 16 * HALF + s]
                b = TV[32 * vb + COL, 16 * HALF + s]
                a = np.nan_to_num(a, nan=123.0); b = np.nan_to_num(b, nan=456.0)   # garbage rows must only reach unused outputs
                acc[ob][vb] = mfma_32x32x2(a.astype(f32), b.astype(f32), acc[ob][vb])
    return acc


def grad_tiles_to_matrix(acc, rows, cols):
    nu, nv = len(acc), len(acc[0])
    G = np.zeros((32 * nu, 32 * nv), f32)
    for ob in range(nu):
        for vb in range(nv):
            for r in range(16):
                G[32 * ob + rowmap(r, HALF), 32 * vb + COL] = acc[ob][vb][r]
    return G[:rows, :cols]


def backward(feats, rgb, W1, W2, W3, v_out, residual=True):
    F = feats.shape[1]
    _, st = forward(feats, rgb, W1, W2, W3, residual)
    x, h1, h2, aff = st["x"], st["h1"], st["h2"], st["aff"]
    v_rgb = rgb_grad(aff, v_out, residual)
    d_aff = affine_grad_tile(v_out, rgb)
    # W3
    g3 = outer_over_pixels(tile_to_lds([d_aff]), tile_to_lds(h2), [[zeros_tile(), zeros_tile()]], 1, 2)
    d_h2 = chained_transposed(W3, [d_aff], 2, k_regs=8)
    d_z2 = [(d * (1 - h * h)).astype(f32) for d, h in zip(d_h2, h2)]
    # W2
    g2 = outer_over_pixels(tile_to_lds(d_z2), tile_to_lds(h1), [[zeros_tile(), zeros_tile()], [zeros_tile(), zeros_tile()]], 2, 2)
    d_h1 = chained_transposed(W2, d_z2, 2)
    d_z1 = [(d * (1 - h * h)).astype(f32) for d, h in zip(d_h1, h1)]
    # W1
    g1 = outer_over_pixels(tile_to_lds(d_z1), features_to_lds(x), [[zeros_tile()], [zeros_tile()]], 2, 1)
    d_x = chained_transposed(W1, d_z1, 1)[0]                     # rows = features (rowmap), cols = pixels
    v_feats = tile_to_matrix(d_x)[:F].T.copy()                   # lane (px, h) stores float4 groups f = 8 g + 4 h + 0..3
    return dict(v_rgb=v_rgb, v_feats=v_feats, v_w1=grad_tiles_to_matrix(g1, HID, F), v_w2=grad_tiles_to_matrix(g2, HID, HID),
                v_w3=grad_tiles_to_matrix(g3, OUTP, HID))


# ---------------------------------------------------------------------------------------------------------------------------------
# the feature slice folded into the head (DESIGN.md "next"): with tiles cut at the grid's cell boundaries the 32 pixels of a tile share
# their (x0, y0) cell on every level, so they touch K = sum_l 4 * gl_l grid "slots" (z, y-corner, x-corner) and
#   features^T [F x 32 px] = G^T [F x K] . Wt [K x 32 px]                 (one more chained D tile in front of layer 1)
#   d(grid slots) [K x F]  += Wt [K x px] . dF [px x F]                   (one more product over the pixels, via the LDS transposes)
#   d(gray) [px]            = sum_ch dF^T[ch][px] * (G^T . dWt/dz)[ch][px] (row-wise dot of two D tiles + one exchange with lane ^ 32)
# Slot order: q = off_l + (z * 2 + yb) * 2 + xb, consumed as k = 2 s + half -> the lane half IS the x corner.
# ---------------------------------------------------------------------------------------------------------------------------------
def slot_table(levels):
    """levels: list of (gl, nch).  Returns [(level, z, yb)] per step s (K/2 steps), channel offsets, K."""
    steps, choff, c = [], [], 0
    for l, (gl, nch) in enumerate(levels):
        choff.append(c); c += nch
        for z in range(gl):
            for yb in (0, 1):
                steps.append((l, z, yb))
    return steps, choff, 2 * len(steps)


def slice_weights(steps, cells, deriv=False, gls=None):
    """B operands of the slice: per step s the lane (px, half = x corner) value Wt[slot][px] = wz(z) * wy(yb) * wx(half), or its
    derivative with respect to the guidance coordinate times (gl - 1) and the interior flag (the chain rule down to the gray value).
    cells: per level dict(fx [32], fy scalar, z0 [32] int, z1 [32] int, fz [32], interior [32] bool)."""
    out = []
    for (l, z, yb) in steps:
        c = cells[l]
        wx = np.where(HALF == 1, c["fx"][COL], 1 - c["fx"][COL])
        wy = c["fy"] if yb else 1 - c["fy"]
        if not deriv:
            wz = (c["z0"][COL] == z) * (1 - c["fz"][COL]) + (c["z1"][COL] == z) * c["fz"][COL]
        else:
            wz = ((c["z1"][COL] == z).astype(f32) - (c["z0"][COL] == z).astype(f32)) * (gls[l] - 1) * c["interior"][COL]
        out.append((wz * wy * wx).astype(f32))
    return out


def grid_operands(steps, choff, levels, region_grids):
    """A operands of the slice: lane (i = channel, half = x corner) holds G_l[slot][ch] of the region's 2 x 2 x gl nodes (zero when
    the channel belongs to another level).  region_grids[l]: [nch, gl, 2, 2] the nodes (z, yb, xb) of this region."""
    out = []
    for (l, z, yb) in steps:
        nch = levels[l][1]
        ch = COL - choff[l]
        ok = (ch >= 0) & (ch < nch)
        out.append(np.where(ok, region_grids[l][np.clip(ch, 0, nch - 1), z, yb, HALF], 0).astype(f32))
    return out


def slice_tile(a_ops, b_ops):
    acc = zeros_tile()
    for a, b in zip(a_ops, b_ops):
        acc = mfma_32x32x2(a, b, acc)
    return acc                                                    # rows = channels, cols = pixels


def layer_first_chained(W1, xt, F):
    """Layer 1 fed by the slice's D tile: step = register r of the tile (channel rowmap(r, half)), only the registers whose row is
    a real channel."""
    out = []
    for o in range(2):
        acc = zeros_tile()
        for r in range(16):
            if 8 * (r >> 2) >= F:                                 # rows 8 g + 4 h + (r & 3): the whole register group is padding
                continue
            k = rowmap(r, HALF)
            a = np.where(k < F, W1[32 * o + COL, np.minimum(k, F - 1)], 0).astype(f32)
            acc = mfma_32x32x2(a, xt[r], acc)
        out.append(acc)
    return out


def fused_forward(levels, region_grids, cells, rgb, W1, W2, W3, residual=True):
    F = sum(n for _, n in levels)
    steps, choff, K = slot_table(levels)
    xt = slice_tile(grid_operands(steps, choff, levels, region_grids), slice_weights(steps, cells))
    h1 = [np.tanh(t).astype(f32) for t in layer_first_chained(W1, xt, F)]
    h2 = [np.tanh(t).astype(f32) for t in layer_chained(W2, h1, 2)]
    aff = layer_chained(W3, h2, 1)[0]
    return apply_affine(aff, rgb, residual), dict(xt=xt, h1=h1, h2=h2, aff=aff, steps=steps, choff=choff, K=K)


def fused_backward(levels, region_grids, cells, rgb, W1, W2, W3, v_out, residual=True):
    F = sum(n for _, n in levels)
    gls = [g for g, _ in levels]
    _, st = fused_forward(levels, region_grids, cells, rgb, W1, W2, W3, residual)
    xt, h1, h2, aff, steps, choff, K = (st[k] for k in ("xt", "h1", "h2", "aff", "steps", "choff", "K"))
    v_rgb = rgb_grad(aff, v_out, residual)
    d_aff = affine_grad_tile(v_out, rgb)
    g3 = outer_over_pixels(tile_to_lds([d_aff]), tile_to_lds(h2), [[zeros_tile(), zeros_tile()]], 1, 2)
    d_h2 = chained_transposed(W3, [d_aff], 2, k_regs=8)
    d_z2 = [(d * (1 - h * h)).astype(f32) for d, h in zip(d_h2, h2)]
    g2 = outer_over_pixels(tile_to_lds(d_z2), tile_to_lds(h1), [[zeros_tile(), zeros_tile()], [zeros_tile(), zeros_tile()]], 2, 2)
    d_h1 = chained_transposed(W2, d_z2, 2)
    d_z1 = [(d * (1 - h * h)).astype(f32) for d, h in zip(d_h1, h1)]
    g1 = outer_over_pixels(tile_to_lds(d_z1), tile_to_lds([xt]), [[zeros_tile()], [zeros_tile()]], 2, 1)   # the slice tile IS in D layout
    d_x = chained_transposed(W1, d_z1, 1)[0]                      # rows = channels
    # guidance: G^T . dWt/dz, dotted with d_x row by row; each half holds half of the rows
    a_ops = grid_operands(steps, choff, levels, region_grids)
    dxdz = slice_tile(a_ops, slice_weights(steps, cells, deriv=True, gls=gls))
    part = (d_x * dxdz).sum(axis=0).astype(f32)                   # over the 16 registers of a lane
    v_gray = (part + part[LANES ^ 32])[:32]
    # grid slots: Wt (lane = pixel) -> LDS rows = slot; d_x tile -> LDS rows = channel; product over the pixels
    wt = slice_weights(steps, cells)
    nblk = (K + 31) // 32
    TW = np.full((32 * nblk, LDS_STRIDE), np.nan, f32)
    for s, w in enumerate(wt):
        TW[2 * s + HALF, COL] = w
    TW[K:, :] = 0                                                 # the kernel zeroes the padding rows of the last block once
    gs = outer_over_pixels(TW, tile_to_lds([d_x]), [[zeros_tile()] for _ in range(nblk)], nblk, 1)
    G = grad_tiles_to_matrix(gs, K, F)                            # [slot, channel]
    v_region = []
    for l, (gl, nch) in enumerate(levels):
        off = 2 * [i for i, s in enumerate(steps) if s[0] == l][0]
        v_region.append(G[off:off + 4 * gl, choff[l]:choff[l] + nch].reshape(gl, 2, 2, nch).transpose(3, 0, 1, 2).copy())
    return dict(v_rgb=v_rgb, v_gray=v_gray, v_w1=grad_tiles_to_matrix(g1, HID, F), v_w2=grad_tiles_to_matrix(g2, HID, HID),
                v_w3=grad_tiles_to_matrix(g3, OUTP, HID), v_region=v_region)

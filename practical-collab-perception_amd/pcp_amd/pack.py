"""Host-side weight preparation for the HIP kernels: BatchNorm folding and repacking into the layouts of
include/pcp_hip.h.  Pure torch (runs on CPU or GPU tensors); done once per set of weights, never per frame.

Layouts (CK = 16 input channels per staged slice):
  conv3x3      W[cout, cin, 3, 3]      -> [cin/16][9 (ky*3+kx)][cout_pad][16]
  plain 1x1    W[cout, cin(,1,1)]      -> [cin/16][cout_pad][16]
  conv k2 s2   W[cout, cin, 2, 2]      -> [(4*cin)/16][cout_pad][16]  with K ordered (tap = ky*2+kx, cin)
  convT k2 s2  W[cin, cout, 2, 2]      -> [cin/16][4*cout_pad][16]    with N ordered (tap = ky*2+kx, cout)
  convT k1 s1  W[cin, cout, 1, 1]      -> plain with W^T
"""
import torch

CK = 16


def round_up(v, m):
    return (v + m - 1) // m * m


def fold_bn(weight, bn_weight, bn_bias, bn_mean, bn_var, eps, conv_bias=None, out_axis=0):
    """y = bn(conv(x) + conv_bias)  ==  conv'(x) + bias' with conv' = conv * s, bias' = (conv_bias - mean) * s + beta,
    s = gamma / sqrt(var + eps).  Computed in float64 and rounded once to float32."""
    s = bn_weight.double() / torch.sqrt(bn_var.double() + eps)
    shape = [1] * weight.dim()
    shape[out_axis] = -1
    w = (weight.double() * s.view(shape)).float()
    b0 = conv_bias.double() if conv_bias is not None else torch.zeros_like(s)
    b = ((b0 - bn_mean.double()) * s + bn_bias.double()).float()
    return w, b


def _pad_rows(mat, n_pad):
    """mat: [n, k] -> [n_pad, k] zero padded"""
    if mat.shape[0] == n_pad:
        return mat
    out = mat.new_zeros((n_pad, mat.shape[1]))
    out[:mat.shape[0]] = mat
    return out


def _slice_k(mat):
    """[n_pad, K] -> [K/16][n_pad][16] contiguous"""
    n, k = mat.shape
    assert k % CK == 0, 'contraction length must be a multiple of 16, got %d' % k
    return mat.view(n, k // CK, CK).permute(1, 0, 2).contiguous()


def pad_bias(bias, n_pad):
    out = bias.new_zeros(n_pad)
    out[:bias.shape[0]] = bias
    return out.contiguous()


def pack_conv3x3(w, bias, n_tile=32):
    """w: [cout, cin, 3, 3] (already BN-folded).  Returns (packed, bias_pad, cout_pad)."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin % CK == 0
    cout_pad = round_up(cout, 64 if cout > 32 else n_tile)
    wp = w.new_zeros((cout_pad, cin, 3, 3))
    wp[:cout] = w
    # [cout_pad, cin/16, 16, 9] -> [cin/16, 9, cout_pad, 16]
    packed = wp.view(cout_pad, cin // CK, CK, 9).permute(1, 3, 0, 2).contiguous()
    return packed, pad_bias(bias, cout_pad), cout_pad


WINO_CK = 8
_WINO_G = [[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]]


def pack_conv3x3_winograd(w, bias):
    """w: [cout, cin, 3, 3] (BN-folded) -> U = G g G^T in float64, rounded once, packed [cin/8][16][cout_pad][8]."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin % WINO_CK == 0
    cout_pad = round_up(cout, 64)
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    u = torch.einsum('ia,ncab,jb->ncij', G, w.double(), G).float()              # [cout, cin, 4, 4]
    up = u.new_zeros((cout_pad, cin, 16))
    up[:cout] = u.reshape(cout, cin, 16)
    packed = up.view(cout_pad, cin // WINO_CK, WINO_CK, 16).permute(1, 3, 0, 2).contiguous()
    return packed, pad_bias(bias, cout_pad), cout_pad


def pack_conv3x3_winograd_ws(w, bias):
    """w: [cout, cin, 3, 3] (BN-folded) -> U = G g G^T in float64, rounded once, in the fragment order of csrc/wino_ws.hip:
    [cin/2 (channel pair)][cout_pad/32 (block)][2 (position half: rows {0,1} | {2,3})][2 (row of the half)][64 lanes = (channel parity h,
    cout r)][4 (position column)] -- a wave's B operand of one channel pair is two contiguous 1-KiB runs."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin % 2 == 0
    cout_pad = round_up(cout, 64)
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    u = torch.einsum('ia,ncab,jb->ncij', G, w.double(), G).float()              # [cout, cin, 4 (i), 4 (j)]
    up = u.new_zeros((cout_pad, cin, 4, 4))
    up[:cout] = u
    x = up.view(cout_pad // 32, 32, cin // 2, 2, 2, 2, 4)                        # (blk, r, kp, h, ph, row, j)
    x = x.permute(2, 0, 4, 5, 3, 1, 6)                                           # (kp, blk, ph, row, h, r, j)
    packed = x.new_zeros((cin // 2 + 2,) + tuple(x.shape[1:]))                   # two zero channel pairs: the kernel's B prefetch overruns
    packed[:cin // 2] = x
    return packed.contiguous(), pad_bias(bias, cout_pad), cout_pad


def unpack_winograd_ws(packed, cout):
    """inverse of pack_conv3x3_winograd_ws -> U [cout, cin, 4, 4] (CPU layout check)"""
    nkp, nblk = packed.shape[0] - 2, packed.shape[1]
    u = packed[:nkp].permute(1, 5, 0, 4, 2, 3, 6).reshape(nblk * 32, nkp * 2, 4, 4)     # (blk, r, kp, h, ph, row, j)
    return u[:cout]


_WINO4_G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
WINO4_CK = 32          # K slice of the batched GEMM
WINO4_BN = 128         # its N tile


def pack_conv3x3_winograd4(w, bias):
    """w: [cout, cin, 3, 3] (BN-folded) -> F(4x4,3x3) filter transform U = G g G^T in float64, rounded once, packed
    [36 (i*6+j)][cout_pad][cin] (K contiguous: the B operand rows of the batched GEMM)."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin % WINO4_CK == 0 and cout % 4 == 0
    cout_pad = round_up(cout, WINO4_BN)
    G = torch.tensor(_WINO4_G, dtype=torch.float64, device=w.device)
    u = torch.einsum('ia,ncab,jb->ijnc', G, w.double(), G).float()              # [6, 6, cout, cin]
    up = u.new_zeros((36, cout_pad, cin))
    up[:, :cout] = u.reshape(36, cout, cin)
    return up.contiguous(), pad_bias(bias, cout_pad), cout_pad


def pack_conv3x3_winograd4f(w, bias):
    """w: [cout, cin, 3, 3] (BN-folded) -> F(4x4,3x3) filter transform U = G g G^T in float64, rounded once, in the fragment order of the
    FUSED kernel csrc/wino4f.hip: [cin/8][36 (i*6+j)][cout_pad][8] (a wave's B operand of one position and slice = one 1-KiB run)."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin % 8 == 0
    cout_pad = round_up(cout, 64)
    G = torch.tensor(_WINO4_G, dtype=torch.float64, device=w.device)
    u = torch.einsum('ia,ncab,jb->ncij', G, w.double(), G).float()              # [cout, cin, 6, 6]
    up = u.new_zeros((cout_pad, cin, 36))
    up[:cout] = u.reshape(cout, cin, 36)
    packed = up.view(cout_pad, cin // 8, 8, 36).permute(1, 3, 0, 2).contiguous()
    return packed, pad_bias(bias, cout_pad), cout_pad


def pack_conv3x3_winograd4h(w, bias):
    """the F(4x4,3x3) filter transform in the fragment order of csrc/wino4h.hip: [cin/8][36][cout_pad/64][64 lanes][8]; lane l = 16 kq + c
    holds, at index 2 nb + ks, U[position][input channel 8 s + 4 ks + kq][output channel 64 n + 16 nb + c] (the A operand of
    v_mfma_f32_16x16x4_f32 for the four 16-channel blocks and the two k steps of a slice: two 16-byte loads per lane)"""
    packed, b, cout_pad = pack_conv3x3_winograd4f(w, bias)                      # [S, 36, cout_pad, 8]
    return repack_winograd4f_to_4h(packed), b, cout_pad


def repack_winograd4f_to_4h(packed):
    """[cin/8][36][cout_pad][8] (k_wino4f's order) -> k_wino4h's lane order (a pure permutation)"""
    S, _p, cout_pad, _k = packed.shape
    v = packed.view(S, 36, cout_pad // 64, 4, 16, 2, 4)                         # [s, pos, n, nb, c, ks, kq]
    return v.permute(0, 1, 2, 6, 4, 3, 5).contiguous().view(S, 36, cout_pad // 64, 64, 8)


def pack_conv3x3_winograd4c(w, bias):
    """the F(4x4,3x3) filter transform in the fragment order of csrc/wino4c.hip: [cin/8][cout_pad/16][18 position pairs][64 lanes][4]; lane
    l = 16 kq + c holds, at index 2 e + ks, U[position 2 q + e][input channel 8 s + 4 ks + kq][output channel 16 g + c] (the A operand of
    v_mfma_f32_16x16x4_f32 for both k steps of two positions: one 16-byte load per lane and position pair)"""
    packed, b, cout_pad = pack_conv3x3_winograd4f(w, bias)                      # [S, 36, cout_pad, 8]
    return repack_winograd4f_to_4c(packed), b, cout_pad


def repack_winograd4f_to_4c(packed):
    """[cin/8][36][cout_pad][8] (k_wino4f's order) -> k_wino4c's order (a pure permutation)"""
    S, _p, cout_pad, _k = packed.shape
    v = packed.view(S, 18, 2, cout_pad // 16, 16, 2, 4)                         # [s, q, e, g, c, ks, kq]
    return v.permute(0, 3, 1, 6, 4, 2, 5).contiguous().view(S, cout_pad // 16, 18, 64, 4)


def pack_conv3x3_sparse_s2(w, bias):
    """w: [cout <= 64, 64, 3, 3] (BN-folded) -> [9 (ky*3+kx)][64][64 (cin)] for pcp_sparse_conv3x3_s2 (rows >= cout zero)."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin == 64 and cout <= 64 and cout % 4 == 0
    wp = w.new_zeros((9, 64, 64), dtype=torch.float32)
    wp[:, :cout] = w.float().permute(2, 3, 0, 1).reshape(9, cout, cin)
    return wp.contiguous(), pad_bias(bias, 64)


def pack_conv3x3_bf16x3(w, bias):
    """w: [cout, cin, 3, 3] (BN-folded) -> split hi = bf16(w), lo = bf16(w - hi), packed bf16 [cin/16][cout_pad/64][hi|lo][9][2][64][8]
    (k-half, cout, 8 channels: the fragment image of v_mfma_f32_32x32x16_bf16).  Returns (packed int16 view, bias_pad, cout_pad)."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin % CK == 0
    cout_pad = round_up(cout, 64)
    wp = w.new_zeros((cout_pad, cin, 9))
    wp[:cout] = w.reshape(cout, cin, 9).float()
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.float()).to(torch.bfloat16)
    x = torch.stack([hi, lo], 0)                                                   # (hl, cout_pad, cin, tap)
    x = x.view(2, cout_pad // 64, 64, cin // CK, 2, 8, 9)                          # (hl, ct, n, s, h, j, tap)
    packed = x.permute(3, 1, 0, 6, 4, 2, 5).contiguous()                           # (s, ct, hl, tap, h, n, j)
    return packed.view(torch.int16), pad_bias(bias, cout_pad), cout_pad


def pack_plain(w, bias):
    """w: [cout, cin] -> ([cin/16][cout_pad][16], bias_pad, cout_pad)"""
    w = w.reshape(w.shape[0], -1)
    cout = w.shape[0]
    cout_pad = round_up(cout, 64 if cout > 32 else 32)
    return _slice_k(_pad_rows(w, cout_pad)), pad_bias(bias, cout_pad), cout_pad


def pack_conv2x2_s2(w, bias):
    """Conv2d(k=2, s=2): w [cout, cin, 2, 2]; K index = (ky*2+kx)*cin + c."""
    cout, cin = w.shape[0], w.shape[1]
    mat = w.permute(0, 2, 3, 1).reshape(cout, 4 * cin)
    cout_pad = round_up(cout, 64 if cout > 32 else 32)
    return _slice_k(_pad_rows(mat, cout_pad)), pad_bias(bias, cout_pad), cout_pad


def pack_convT2x2_s2(w, bias):
    """ConvTranspose2d(k=2, s=2): w [cin, cout, 2, 2]; N index = (ky*2+kx)*cout_pad + co."""
    cin, cout = w.shape[0], w.shape[1]
    cout_pad = round_up(cout, 64 if cout > 32 else 32)
    mat = w.new_zeros((4, cout_pad, cin))
    mat[:, :cout] = w.permute(2, 3, 1, 0).reshape(4, cout, cin)
    return _slice_k(mat.view(4 * cout_pad, cin)), pad_bias(bias, cout_pad), cout_pad


def pack_convT1x1(w, bias):
    """ConvTranspose2d(k=1, s=1): w [cin, cout, 1, 1] == 1x1 conv with the transposed matrix."""
    return pack_plain(w[:, :, 0, 0].t().contiguous(), bias)

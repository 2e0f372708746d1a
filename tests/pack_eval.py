"""Plain-torch evaluators / inverses of the packed weight forms of pcp_amd/pack.py -- TEST INFRASTRUCTURE (moved out of the product package in
round 6): they validate the Winograd transform matrices and the packed layouts on the CPU (tests/test_host_cpu.py); nothing in the product
path evaluates a convolution in torch."""
import torch

from pcp_amd.pack import CK, WINO_CK


def winograd4_reference(x, packed, bias, cout):
    """Plain-torch evaluation of the packed F(4x4,3x3) form with the transforms of csrc/wino4.hip (validates matrices + layout on the
    CPU): x (B, cin, H, W), H, W multiples of 4."""
    _p36, cout_pad, cin = packed.shape
    u = packed.view(6, 6, cout_pad, cin)[:, :, :cout]                               # [i, j, cout, cin]
    B, _, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                       [0, 4, 0, -5, 0, 1]], dtype=x.dtype)
    AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=x.dtype)
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                                        # [B, cin, H/4, W/4, 6, 6]
    v = torch.einsum('ia,bcyxae,je->bcyxij', BT, d, BT)
    m = torch.einsum('ijnc,bcyxij->bnyxij', u.to(x.dtype), v)
    y = torch.einsum('ki,bnyxij,lj->bnyxkl', AT, m, AT)                          # [B, cout, H/4, W/4, 4, 4]
    out = y.permute(0, 1, 2, 4, 3, 5).reshape(B, cout, H, W)
    return out + bias[:cout].to(x.dtype).view(1, -1, 1, 1)


def winograd_reference(x, packed, bias, cout):
    """Plain-torch evaluation of the packed Winograd form (validates transforms + layout on the CPU): x (B, cin, H, W), H, W even."""
    nsl, _sixteen, cout_pad, _ck = packed.shape
    cin = nsl * WINO_CK
    u = packed.permute(2, 0, 3, 1).reshape(cout_pad, cin, 4, 4)[:cout]           # [cout, cin, i, j]
    B, _, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=x.dtype)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=x.dtype)
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                        # [B, cin, H/2, W/2, 4, 4]
    v = torch.einsum('ia,bcyxae,je->bcyxij', BT, d, BT)
    m = torch.einsum('ncij,bcyxij->bnyxij', u, v)
    y = torch.einsum('ia,bnyxae,je->bnyxij', AT, m, AT)                           # [B, cout, H/2, W/2, 2, 2]
    out = y.permute(0, 1, 2, 4, 3, 5).reshape(B, cout, H, W)
    return out + bias[:cout].view(1, -1, 1, 1)


def unpack_conv3x3(packed, cout, cin):
    nsl, _nine, cout_pad, _ck = packed.shape
    w = packed.permute(2, 0, 3, 1).reshape(cout_pad, nsl * CK, 3, 3)
    return w[:cout, :cin]


def unpack_plain(packed, cout):
    nsl, n_pad, _ck = packed.shape
    return packed.permute(1, 0, 2).reshape(n_pad, nsl * CK)[:cout]

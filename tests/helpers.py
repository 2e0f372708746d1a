"""Shared test helpers: golden loading, arch/state reconstruction (no reference access at run time)."""
import json
import os

import numpy as np

from pcp_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if 'meta_json' in d:
        d['meta'] = json.loads(str(d.pop('meta_json')))
    return d


def arch_of(meta):
    from oracle import model as omodel
    return omodel.arch_from_cfg(meta['model'], meta['pc_range'], meta['voxel_size'])


def match_boxes(a_boxes, a_scores, b_boxes, b_scores, tol=1e-3):
    """Greedy one-to-one matching by score then centre distance; returns (n_matched, max_abs_diff)."""
    used = np.zeros(b_boxes.shape[0], bool)
    worst = 0.0
    n = 0
    for i in range(a_boxes.shape[0]):
        d = np.abs(b_boxes[:, :2] - a_boxes[i, :2]).sum(1) + np.abs(b_scores - a_scores[i]) + used * 1e9
        if d.size == 0:
            break
        j = int(np.argmin(d))
        diff = np.abs(b_boxes[j] - a_boxes[i])
        diff[6] = min(diff[6], abs(diff[6] - 2 * np.pi))
        e = max(float(diff.max()), float(abs(b_scores[j] - a_scores[i])))
        if e <= tol:
            used[j] = True
            n += 1
            worst = max(worst, e)
    return n, worst


def assert_same_final_set(ref_boxes, ref_scores, got_boxes, got_scores, tol=1e-5):
    """the two detection lists hold the SAME boxes: equal count and a one-to-one match of every box (7 parameters and score) within
    `tol` -- the order among exactly tied scores is implementation defined (torch.topk), so lists are compared as sets"""
    assert got_boxes.shape[0] == ref_boxes.shape[0], (got_boxes.shape[0], ref_boxes.shape[0])
    n, worst = match_boxes(ref_boxes, ref_scores, got_boxes, got_scores, tol=tol)
    assert n == ref_boxes.shape[0], (n, ref_boxes.shape[0], worst)


def assert_subset_of_candidates(got_boxes, got_scores, cand_boxes, cand_scores, tol=1e-3):
    """every detection is (within tol) one of the reference's own NMS-input candidates"""
    n, worst = match_boxes(got_boxes, got_scores, cand_boxes, cand_scores, tol=tol)
    assert n == got_boxes.shape[0], (n, got_boxes.shape[0], worst)


def postprocess_reference_maps(model, maps):
    """decode + rotated NMS + gather of the product's CenterHead run on GIVEN head maps ({name: (B, k, H, W) numpy}, e.g. the
    reference's own): isolates the integer / selection stage of the path from the float noise of the conv stack in front of it"""
    import torch
    head = model.dense_head
    pk = head.packed()
    entry = pk['heads'][0]
    B, _, H, W = maps['hm'].shape
    ld = int(entry['offs'][-1])
    ld_pad = (ld + 3) // 4 * 4
    buf = torch.zeros((B, H, W, ld_pad), dtype=torch.float32, device='cuda')
    for i, name in enumerate(entry['names']):
        buf[..., int(entry['offs'][i]):int(entry['offs'][i + 1])] = torch.from_numpy(maps[name]).cuda().permute(0, 2, 3, 1)
    return head.generate_predicted_boxes(B, [buf.contiguous()], pk)

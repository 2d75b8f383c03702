"""CPU checks of the Swin host logic (no GPU, no HIP library calls): the engine's window helpers against the
oracle's restatement of swin_quant.py:18-50, 73-88, 223-249."""
import numpy as np

from oracle import oracle as orc

import ivit_amd  # noqa: F401
from ivit_amd.swin_engine import rel_position_index, shift_mask_regions, window_row_map


def test_host_window_helpers_match_oracle():
    for ws in (2, 4, 7):
        assert np.array_equal(rel_position_index(ws), orc.swin_rel_index(ws))
    for H, ws in ((56, 7), (28, 7), (14, 7), (8, 4)):
        reg = shift_mask_regions(H, H, ws, ws // 2)
        mask = np.where(reg[:, :, None] != reg[:, None, :], np.float32(-100.0), np.float32(0.0))
        # oracle: mask[w, i, j] from mw[:, None, :] - mw[:, :, None] -> symmetric in (i, j)
        assert np.array_equal(mask, orc.swin_attn_mask(H, H, ws, ws // 2))
    # the row map restates roll + window_partition
    B, H, ws, shift, C = 2, 14, 7, 3, 5
    x = np.arange(B * H * H * C, dtype=np.int32).reshape(B, H, H, C)
    part = orc._win_partition(np.roll(x, (-shift, -shift), axis=(1, 2)), ws).reshape(-1, C)
    rm = window_row_map(B, H, H, ws, shift)
    out = np.empty_like(part)
    out[rm] = x.reshape(-1, C)
    assert np.array_equal(out, part)



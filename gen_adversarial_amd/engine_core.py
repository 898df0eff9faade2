"""Shared pieces of the plan builder (engine.py and its per-network builder mixins): constants, the conv tuning
table, the weight store and the activation record."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

RES_SCALE = 0.1          # `0.1 * self.residual(x)` — architecture.py:133,183
IMG_LD = 8                  # channel pitch of the NHWC image tensors (3 channels + zero padding)
WS_FLOATS = 32 * 1024 * 1024     # split-K workspace shared by every conv of an engine (128 MB)
TUNE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_tune_gfx950.json')
_TUNE_CACHE: Optional[dict] = None


def tune_cache() -> dict:
    """(tile, splits) per conv shape, measured on an MI355X by Engine.autotune and kept in-tree."""
    global _TUNE_CACHE
    if _TUNE_CACHE is None:
        _TUNE_CACHE = {}
        if os.path.exists(TUNE_FILE):
            with open(TUNE_FILE) as f:
                _TUNE_CACHE = json.load(f)
    return _TUNE_CACHE


def conv_key(d) -> str:
    return ('b3_' if d.w_hi else '') + '_'.join(str(int(v)) for v in (
        d.N * d.Ho * d.Wo, d.Cout, d.C1, d.C2, d.KH, d.sn, d.sd, d.Hi, d.pro_act, bool(d.pro_scale), d.pro_per_row,
        bool(d.dact_x), d.dact_act, bool(d.addend), bool(d.addend2), d.addend_bcast_n)) + (
        f'_kw{d.KW}' if d.KW != d.KH else '')


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class WeightStore:
    """Folded weights on the device, shared by every engine (row count) built for one model."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cache: Dict[str, dict] = {}
        self.splits: Dict[int, tuple] = {}
        self.bytes = 0

    def split(self, w: torch.Tensor):
        """bf16 (hi, lo) pair of a device weight tensor, hi = bf16(w), lo = bf16(w - hi); made once per tensor."""
        k = w.data_ptr()
        if k not in self.splits:
            hi = w.to(torch.bfloat16)
            lo = (w - hi.float()).to(torch.bfloat16)
            self.splits[k] = (hi.contiguous(), lo.contiguous(), w)
            self.bytes += 4 * w.numel()
        return self.splits[k][0], self.splits[k][1]

    def frag3(self, w: torch.Tensor, m16: bool = False) -> torch.Tensor:
        """the split weights of a 3x3 conv ([Cout][9 * C] fp32, C % 32 == 0) in the MFMA-fragment order of ga_conv_desc.w_frag
        (bf16 [ceil(Cout/128)][C/32][9][4 waves][2 k steps][hi | lo][64 lanes][8]); made once per tensor.
        m16: the order of the 16x16x32 fragments, [..][4 waves][2 halves of the wave's 32 channels][hi | lo][64 lanes][8], lane =
        16 * (k octet of the 32-deep chunk) + channel"""
        k = ('frag3m16' if m16 else 'frag3', w.data_ptr())
        if k not in self.splits:
            hi, lo = self.split(w)
            cout, kk = w.shape
            c = kk // 9
            nt, nkc = (cout + 127) // 128, c // 32

            def arr(t):
                tp = torch.zeros(nt * 128, kk, dtype=torch.bfloat16, device=t.device)
                tp[:cout] = t
                if m16:     # [nt, wave, half, channel, tap, chunk, k octet, e] -> [nt, chunk, tap, wave, half, k octet, channel, e]
                    return tp.view(nt, 4, 2, 16, 9, nkc, 4, 8).permute(0, 5, 4, 1, 2, 6, 3, 7)
                # [nt, wave, row, tap, chunk, k step, lane half, e] -> [nt, chunk, tap, wave, k step, lane half, row, e]
                return tp.view(nt, 4, 32, 9, nkc, 2, 2, 8).permute(0, 4, 3, 1, 5, 6, 2, 7)
            f = torch.stack([arr(hi), arr(lo)], dim=5).contiguous()          # hi | lo between the k step / half and the lane
            self.splits[k] = (f, w)
            self.bytes += 2 * f.numel()
        return self.splits[k][0]

    def frag_thin(self, w: torch.Tensor) -> torch.Tensor:
        """the split weights of a 3x3 conv with C = 32 or 64 input channels ([Cout][9 * C] fp32) in the order of tile 11 (conv_thin3.hip):
        bf16 [ceil(Cout/32)][9 taps][C/16 k steps][hi | lo][64 lanes][8], lane = 32 * (k octet of the 16-deep step) + channel"""
        k = ('frag_thin', w.data_ptr())
        if k not in self.splits:
            hi, lo = self.split(w)
            cout, kk = w.shape
            assert kk in (9 * 32, 9 * 64), tuple(w.shape)
            nt, ks = (cout + 31) // 32, kk // 9 // 16

            def arr(t):
                tp = torch.zeros(nt * 32, kk, dtype=torch.bfloat16, device=t.device)
                tp[:cout] = t
                # [nt, channel, tap, k step, k octet, e] -> [nt, tap, k step, k octet, channel, e]
                return tp.view(nt, 32, 9, ks, 2, 8).permute(0, 2, 3, 4, 1, 5)
            f = torch.stack([arr(hi), arr(lo)], dim=3).contiguous()            # hi | lo between the k step and the lane
            self.splits[k] = (f, w)
            self.bytes += 2 * f.numel()
        return self.splits[k][0]

    def get(self, key: str, fn):
        if key not in self.cache:
            d = {k: v.to(self.device, dtype=torch.float32).contiguous() for k, v in fn().items()}
            self.bytes += sum(v.numel() * 4 for v in d.values())
            self.cache[key] = d
        return self.cache[key]


class Act:
    """An NHWC activation buffer plus its (lazily allocated) gradient buffer.  With K cotangents per forward row
    (Engine(cot_rep=K): the K-cotangent backward plan) the gradient buffer has n * K rows, cotangent k of forward row r at row
    r * K + k; the backward ops read the saved activation of cotangent row m at row m // K (`act_rep` / `dact_rep` of ga_ops.h)."""

    def __init__(self, eng: "Engine", n, h, w, c, name=''):
        self.eng, self.n, self.h, self.w, self.c, self.name = eng, n, h, w, c, name
        self.t = eng.alloc((n, h, w, c))
        self._g = None
        self.g_written = False
        eng.acts[name] = self

    @property
    def g(self) -> torch.Tensor:
        if self._g is None:
            self._g = self.eng.alloc((self.n * getattr(self.eng, 'cot_rep', 1), self.h, self.w, self.c))
        return self._g



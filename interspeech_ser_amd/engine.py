"""Host-side encoders: the drop-in for the reference's model call.

The reference does (preprocess_speech.py:49-50,66-67 / preprocess_whisper.py:57,71)

    hidden_states = model(**inputs, output_hidden_states=True).hidden_states
    hidden_states = model.encoder(input_features, output_hidden_states=True).hidden_states

one utterance at a time.  ``SpeechEncoder.forward`` / ``WhisperEncoder.forward``
take a *ragged batch* of raw waveforms and return the same L+1 hidden states for
every utterance, computed entirely by libserhip's gfx950 kernels.  PyTorch is used
for device memory, H2D/D2H copies and the stream handle only.

HBM layout: utterances are packed, never padded -- a batch is one ``[rows, C]``
matrix and ``frame_offs[b]`` is utterance b's first row.  Packed rows make a batched
run arithmetically identical to the reference's batch-of-one loop (no padding frames
exist, so no attention / conv masking is needed) and waste no FLOPs on padding.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import GemmArgs, check, lib
from .config import EncoderGeometry, FAMILY_ROBERTA, FAMILY_WAVLM, FAMILY_WHISPER

MODES = {"bf16": _lib.MODE_BF16, "fp32x": _lib.MODE_FP32X, "f16": _lib.MODE_FP16, "f16q": _lib.MODE_FP16, "f16a": _lib.MODE_FP16,
         "f16x": _lib.MODE_FP16X, "f16m": _lib.MODE_FP16M, "f16mf": _lib.MODE_FP16M}
# "f16mf" (round 5, the drivers' default): "f16m" where it is benign -- FC1 and FC2 (2/3 of the layer FLOPs) in every layer, the packed
# projection from a third of the depth on (_EncoderBase.qkv_m_from, _lay_modes); the conv stem, ser_attention, the output projection and
# the first third's packed projections keep "f16x"'s three products on fp16 hi + lo planes.  oracle/numerics_whatif_f16m.py (site lists): under sharp
# attention the error of "f16m" comes from the packed projections of the FIRST layers (an error injected by layer i passes through L - i
# more softmax layers): all of them 4.3e-4 of f16m's 5.2e-4 at 24 layers, from layer 8 on nothing measurable; FC1 + FC2 alone give 1.4e-4
# (sharp x2), 2.7e-5 (LoRA), 1.3e-5 (plain) -- inside "fp32x"'s on each.  The stem in that format: 1.4e-3 (it stays on 22 bits).
# "f16m" (round 5): the encoder layers' GEMMs on SER_MODE_FP16M operands -- fp16 main product + block-scaled e4m3 cross terms on gfx950's
# v_mfma_scale_f32_16x16x128_f8f6f4: 2 product-equivalents per algorithmic FLOP instead of "f16x"'s 3 (include/ser_hip.h).  The packed
# projection, FC1 and FC2 multiply in it (opt-in, SER_F16M_OUT_M=1, head dim 64: the output projection too, on FP16M context rows out of
# ser_attention, ABI 14); ser_attention, the output projection and the conv stem stay on fp16 hi + lo planes.  Operand error ~2^-15 (between "f16"'s 2^-11 and
# "f16x"'s 2^-22): oracle/numerics_whatif_f16m.py, tests/test_gpu_depth.py.
# "f16x" (round 4): the 3-product split EVERYWHERE, like "fp32x", on fp16 hi + lo planes -- 22-bit operands instead of the 16 of the
# bf16 pair at the same cost.  The widest margin of all modes where |values| stay inside fp16's range (65 504).
# "f16": encoder layers on single-product fp16 operands (11 significand bits at the bf16 MFMA rate), the convolutional
# stem -- where operand rounding hurts most and only 13 % of the FLOPs live -- on the 3-product FP32X split.
# "f16q": "f16" with the LOGIT path of every layer fp32-grade: the q / k (+ WavLM gate) columns of the packed projection and
# S = K Q^T inside the attention kernel run the 3-product split on fp16 hi + lo planes (SER_MODE_FP16X), v / P V / output
# projection / feed-forward stay single-product fp16.  A softmax weight moves by (logit error) * ln 2, so q / k rounding is
# what sharp attention maps (trained checkpoints, LoRA-scaled query projections) amplify; ~19 % of the layer FLOPs pay 3x.
# "f16a": the whole ATTENTION BLOCK of every layer (packed projection, attention, output projection) on the fp16 hi + lo split,
# the feed-forward pair (62 % of the layer FLOPs) on single fp16 products.  oracle/numerics_whatif.py: under sharp attention the
# error comes from the attention block as a whole -- rounding v, P, the context rows or the output-projection weights once is
# amplified by the following layers' softmax as much as rounding q and k -- while the feed-forward rounding is benign.
_PLANES = {_lib.MODE_BF16: 1, _lib.MODE_FP32X: 2, _lib.MODE_FP16: 1, _lib.MODE_FP16X: 2, _lib.MODE_FP16M: 2}
_DTYPE = {_lib.MODE_BF16: torch.bfloat16, _lib.MODE_FP32X: torch.bfloat16, _lib.MODE_FP16: torch.float16,
          _lib.MODE_FP16X: torch.float16, _lib.MODE_FP16M: torch.float16}
# MFMA products per algorithmic FLOP of a GEMM launch in each operand format (FP16M: the scaled e4m3 instruction runs at twice the fp16 rate)
_PRODUCTS = {_lib.MODE_BF16: 1, _lib.MODE_FP32X: 3, _lib.MODE_FP16: 1, _lib.MODE_FP16X: 3, _lib.MODE_FP16M: 2}


# A/B knob (tools/): SER_NO_SHIFT=1 turns the shifted operand copy of the encoder layers off (state 0 is still centred)
import os as _os
_NO_SHIFT = _os.environ.get("SER_NO_SHIFT", "0") == "1"
# (Round 3's SER_SPLIT_GATE -- the 2H gate columns as their own narrow launch -- measured -0.3 % and was removed in round 4: DESIGN.md section 10.)
# WavLM's gate pre-activations: computed by ser_attention from the layer input's operand copy (ser_attention_args.gate_x; default), or
# 2H extra columns of the packed projection (SER_GATE_IN_ATTN=0, rounds 1-3: a 13th 256-wide column tile for 32 columns).  Measured on the
# step (tools/gate_in_attn_ab.sh, two A/B pairs per build, one box; profiles/r03_gate_in_attn_ab.txt): bf16 2 018 / 2 018 -> 2 029 / 2 029
# utt/s, f16a 1 134 / 1 135 -> 1 154 / 1 159 (there the 13th tile column is a 3-product one).  The first form of the kernel side guarded its
# loads (a branch and a vmcnt(0) each in hipcc's output) and LOST 0.4 % in bf16: the prologue of ser_attention is latency-bound.
# (Read when an encoder is built: _EncoderBase.gate_in_attn.)
# A/B knob: SER_STEM_F16X=0 puts the stem of the f16 / f16q / f16a modes back on bf16 hi + lo planes (rounds 2 / early 3)
_STEM_F16X = _os.environ.get("SER_STEM_F16X", "1") == "1"


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class Sz(int):
    """A per-batch size (row count, utterances, longest utterance) that remembers its name.  It is an ``int`` everywhere;
    a launch recorded into a command list keeps the name so the field can be patched for the next batch."""

    def __new__(cls, value: int, name: str):
        o = int.__new__(cls, value)
        o.name = name
        return o


class Tape:
    """The launches of one forward over a slot's arena, recorded as ``ser_cmd`` entries (include/ser_hip.h).  Pointers
    in an arena never change, so a later batch only patches its sizes (``Sz`` fields) and replays everything with one
    ``ser_run`` call: ~150 Python -> C transitions per batch become one."""

    def __init__(self, capacity: int = 640):
        self.cmds = (_lib.Cmd * capacity)()
        self.n = 0
        self.patches = []                    # (struct view inside cmds, field name, size name)
        self.inputs: Dict[str, tuple] = {}   # named per-batch pointers, e.g. the uploaded waveform
        self.marks: Dict[int, int] = {}      # hidden state index -> number of leading commands that produce states[0..index]
        self._failed = C.c_int32(-1)

    def slot(self, union_field: str):
        if self.n >= len(self.cmds):
            raise _lib.SerHipError("command list capacity exceeded")
        return getattr(self.cmds[self.n].u, union_field)

    def commit(self, op: int, view, **sizes) -> None:
        self.cmds[self.n].op = op
        for field, value in sizes.items():
            if isinstance(value, Sz):
                self.patches.append((view, field, value.name))
        self.n += 1

    def subset(self, keep) -> "Tape":
        """Measurement aid (bench.py ``step_decomposition``): a command list holding copies of the recorded launches whose op
        satisfies ``keep(op)``, with the sizes of the batch that ran last.  Same pointers, same shapes, same kernels -- replayed
        on its own it shows what ONE class of kernels (the GEMMs, the attention kernels, the row kernels) costs in the regime
        the step runs them in (two utterance groups on parallel graph branches)."""
        picked = [i for i in range(self.n) if keep(self.cmds[i].op)]
        t = Tape(max(1, len(picked)))
        for j, i in enumerate(picked):
            C.memmove(C.byref(t.cmds[j]), C.byref(self.cmds[i]), C.sizeof(_lib.Cmd))
        t.n = len(picked)
        return t

    def run(self, sizes: Dict[str, int], stream: int, last_state: Optional[int] = None) -> None:
        """Replay the list; with ``last_state`` only the leading commands that produce hidden states 0..last_state."""
        for view, field, name in self.patches:
            setattr(view, field, sizes[name])
        n = self.n if last_state is None else self.marks.get(last_state, self.n)
        rc = lib.ser_run(self.cmds, n, C.byref(self._failed), stream)
        if rc != 0:
            check(rc, f"ser_run (command {self._failed.value} of {n})")


def _on_stream(fn):
    """Look the launch stream up once for the whole forward (see ``_EncoderBase._s``)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        prev = self._st
        self._st = _stream()
        try:
            return fn(self, *a, **k)
        finally:
            self._st = prev
    return wrapped


class Act:
    """16-bit GEMM operand with 1 (bf16 / fp16) or 2 (bf16 hi/lo) planes: tensor [planes, rows, cols]."""

    def __init__(self, rows: int, cols: int, planes: int, device, zero: bool = False, extra_rows: int = 0,
                 dtype=torch.bfloat16, mx: bool = False):
        alloc = torch.zeros if zero else torch.empty
        self.t = alloc((planes, rows + extra_rows, cols), dtype=dtype, device=device)
        self.rows, self.cols, self.planes = rows, cols, planes
        self.plane_stride = (rows + extra_rows) * cols
        # SER_MODE_FP16M: plane 1 holds the e4m3 cross-term bytes; one uint32 of four block-scale codes per (64-column tile, row)
        self.scale = torch.zeros((cols // 64, rows + extra_rows), dtype=torch.int32, device=device) if mx else None
        self.scale_ld = rows + extra_rows

    @property
    def ptr(self) -> int:
        return self.t.data_ptr()

    def float(self) -> torch.Tensor:
        """fp32 view for tests (hi + lo)."""
        if self.scale is not None:
            raise NotImplementedError("FP16M tensors: plane 1 is not a 16-bit lo plane")
        return self.t[:, : self.rows].float().sum(dim=0)


@dataclass
class Linear:
    w: torch.Tensor            # [planes, N, K] bf16
    b: Optional[torch.Tensor]  # [N] fp32
    N: int
    K: int
    colsum: Optional[torch.Tensor] = None   # [N] fp32: sum_k of the stored (gamma-folded) planes, deferred LayerNorm
    wscale: Optional[torch.Tensor] = None   # SER_MODE_FP16M: [K / 64, N] int32 block-scale words


class HiddenStates:
    """What ``.hidden_states`` is in the reference: L+1 states per utterance.
    ``states`` is one fp32 tensor [L+1, rows, D]; utterance b owns rows
    frame_offs[b]:frame_offs[b+1] (crop to the true frame count is implicit)."""

    def __init__(self, states: torch.Tensor, frame_offs: Sequence[int], computed: Optional[int] = None):
        self.states = states
        self.frame_offs = list(int(x) for x in frame_offs)
        # states[0 .. computed-1] hold this batch's results; a forward stopped early (``last_state``) leaves the rest stale
        self.computed = states.shape[0] if computed is None else int(computed)

    def __len__(self) -> int:
        return self.states.shape[0]

    @property
    def batch(self) -> int:
        return len(self.frame_offs) - 1

    def utterance(self, b: int, layer: int) -> torch.Tensor:
        idx = layer if layer >= 0 else layer + self.states.shape[0]
        if not 0 <= idx < self.computed:
            raise IndexError(f"hidden state {layer} is not available (states 0..{self.computed - 1} were computed)")
        return self.states[idx, self.frame_offs[b]: self.frame_offs[b + 1]]

    def frames(self, b: int) -> int:
        return self.frame_offs[b + 1] - self.frame_offs[b]

    # fp16 range guard (round 5): every kernel that rounds a value to an fp16 operand plane (GEMM / row-kernel epilogues in the f16x, f16m,
    # f16a, f16q, f16 modes) ORs into the slot's device word -- bit 0: a value beyond +-65504 (it saturated), bit 1: beyond half of that.
    # ``range_flag`` is that word (int32 [1] on the device, None in the bf16-plane modes); the caller reads it back with the features
    # and clears it (``take_range_bits``).  Round 4 reduced max|hidden state| with a torch pass on sampled batches only.
    range_flag: Optional[torch.Tensor] = None

    def take_range_bits(self, pinned_out: Optional[torch.Tensor] = None):
        """Enqueue (current stream) the read-back of the guard word into ``pinned_out`` (int32 [1], page-locked) and its reset; with no
        buffer given: synchronous, returns the bits.  0 in the modes that keep no fp16 planes."""
        if self.range_flag is None:
            if pinned_out is not None:
                pinned_out.zero_()
            return 0
        if pinned_out is not None:
            pinned_out.copy_(self.range_flag, non_blocking=True)
            self.range_flag.zero_()
            return None
        bits = int(self.range_flag.item())
        self.range_flag.zero_()
        return bits


class _EncoderBase:
    def __init__(self, geo: EncoderGeometry, device, mode: str):
        if mode not in MODES:
            raise ValueError(f"mode must be one of {list(MODES)}")
        if not torch.cuda.is_available():
            raise _lib.SerHipError("no HIP device visible: the extraction path has no CPU fallback")
        self.geo = geo
        self.device = torch.device(device)
        self.mode_name = mode
        self.mode = MODES[mode]                                   # encoder layers
        # conv stem (+ projection, positional conv): the 3-product split in every parity mode.  The fp16-layer modes take it on fp16 hi + lo
        # planes (22-bit operands, round 3) rather than bf16 hi + lo (16-bit, the "fp32x" mode's): same cost, and the stem's share of the error
        # -- which sharp attention amplifies like any other -- drops by the 6 extra bits per operand
        self.stem_mode = (_lib.MODE_FP16X if _STEM_F16X else _lib.MODE_FP32X) if mode in ("f16", "f16q", "f16a") else self.mode
        if mode in ("f16m", "f16mf"):
            self.stem_mode = _lib.MODE_FP16X
        # WavLM gate inside ser_attention (see the note at the top) -- except beside a packed projection in SER_MODE_FP16M, where it rides as 2H
        # extra output columns (per layer: _lay_modes)
        self.gate_in_attn = _os.environ.get("SER_GATE_IN_ATTN", "1") == "1"
        self.qk_mode = _lib.MODE_FP16X if mode == "f16q" else None             # logit path on its own launch (None: one packed launch)
        self.attn_mode = _lib.MODE_FP16X if mode in ("f16a", "f16m", "f16mf") else self.mode   # attention kernel, context rows, output projection
        self.qkv_mode = self.attn_mode                                         # packed projection (layers before qkv_m_from)
        self.x_mode = self.qk_mode or self.qkv_mode                            # format of the operand copy the packed projection reads
        self.qkv_out_mode = self.x_mode                                        # format of q, k, v (what ser_attention reads)
        # First layer whose PACKED PROJECTION multiplies in SER_MODE_FP16M (see _lay_modes): "f16m" 0 = every layer; "f16mf" a third of the
        # depth (8 of 24, 16 of 48, 11 of 32).  An operand error injected by layer i passes through L - i more softmax layers: the what-if
        # (oracle/numerics_whatif_f16m.py, sites "qkv>=N") puts the packed projection in that format from layer 8 of 24 on at 1.44e-4 / 3.7e-5
        # (sharp x2 / LoRA) against 1.43e-4 / 2.7e-5 with none and 5.2e-4 / 4.7e-4 with all -- its error lives in the first layers.
        self.qkv_m_from: Optional[int] = {"f16m": 0, "f16mf": (geo.num_layers + 2) // 3}.get(mode)
        if mode == "f16mf" and _os.environ.get("SER_F16MF_QKV_FROM"):          # A/B knob (tools/): -1 = never
            v = int(_os.environ["SER_F16MF_QKV_FROM"])
            self.qkv_m_from = None if v < 0 else v
        self.planes, self.stem_planes = _PLANES[self.mode], _PLANES[self.stem_mode]
        self._cache: Dict = {}
        # when a list, every ser_gemm launch appends (start_event, end_event, algorithmic_flops):
        # bench.py uses it for the live roofline figure of the dominant kernel
        self.gemm_trace: Optional[list] = None
        # when a list, every encoder layer appends (start_event, end_event, utterances) around its attention block
        # (packed QKV projection -> attention -> output projection): bench.py's "attention_block" figure
        self.block_trace: Optional[list] = None
        # operand copies with fp16's range: the guard word is live (SER_NO_RANGE_GUARD=1: A/B knob for tools/, never the drivers)
        self.fp16_planes = mode in ("f16x", "f16m", "f16mf", "f16a", "f16q", "f16") and _os.environ.get("SER_NO_RANGE_GUARD", "0") != "1"
        self._flag: Optional[int] = None    # device address of the range-guard word of the slot being launched / recorded
        self._st: Optional[int] = None      # launch stream of the forward in progress (looked up once per forward)
        self._rec: Optional[Tape] = None    # when set, the launch helpers record into it instead of launching

    def _guard_word(self, pl) -> Optional[torch.Tensor]:
        """the slot's range-guard word (allocated with the plan); launch helpers pick its address up from ``self._flag``"""
        if not self.fp16_planes:
            self._flag = None
            return None
        t = pl.get("range_flag")
        if t is None:
            t = pl["range_flag"] = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._flag = t.data_ptr()
        return t

    def _lay_modes(self, i: int) -> dict:
        """Formats around layer i's packed projection: the GEMM's mode, the layer input's operand copy it reads (written by ser_row_center for
        layer 0, by FC2 of layer i - 1 otherwise), q / k / v as ser_attention reads them, and where the WavLM gate is evaluated.  With the
        projection in SER_MODE_FP16M the gate rides as 2H extra output columns: the in-kernel form multiplies the layer input's operand copy,
        whose second plane is e4m3 bytes in that format."""
        if self.qkv_m_from is not None and i >= self.qkv_m_from:
            # ... and with head dim 64 (a head = one 64-column tile) ser_attention can write its context rows as FP16M operands (ABI 14), so that
            # the OUTPUT projection of these layers multiplies in the format too (what-if "qkv>=8+out>=8": 1.44e-4 / 4.1e-5; + 1.6 % on the
            # default's step, same envelope on WavLM-large).  OPT-IN (SER_F16M_OUT_M=1), not the default: bench.py's Whisper-large-v3 record
            # (two 8 x 30 s groups captured as concurrent graph branches, then replayed command lists) came back 0.57 from the oracle with it
            # while the eager single-slot forward of the same batch is right (1.9e-5) -- found with no GPU time left in round 5 to chase it.
            return dict(qkv_mode=_lib.MODE_FP16M, x_mode=_lib.MODE_FP16M, qkv_out_mode=self.attn_mode, gate_in_attn=False,
                        out_m=self.geo.head_dim == 64 and _os.environ.get("SER_F16M_OUT_M", "0") == "1")
        return dict(qkv_mode=self.qkv_mode, x_mode=self.x_mode, qkv_out_mode=self.qkv_out_mode, gate_in_attn=self.gate_in_attn, out_m=False)

    def _check_last_state(self, last_state: Optional[int]) -> Optional[int]:
        if last_state is None:
            return None
        L = self.geo.num_layers
        last_state = int(last_state)
        if not 0 <= last_state <= L:
            raise IndexError("tuple index out of range")         # what hidden_states[N] raises in the reference
        return None if last_state == L else last_state

    def _s(self) -> int:
        """HIP stream of the current forward: torch.cuda.current_stream() costs ~15 us and a forward makes ~150 launches."""
        return self._st if self._st is not None else _stream()

    # ------------------------------------------------------------------ weights
    def _dev_f32(self, t: torch.Tensor) -> torch.Tensor:
        """fp32 copy on the device that the encoder OWNS: a state dict that arrived by broadcast is made of views into one flat
        bucket in HBM (dist.broadcast_state_dict); keeping such a view would pin the whole fp32 checkpoint (1.3 GB WavLM-large,
        8.6 GB XLS-R-2B) beside the 16-bit planes for the life of the run, and its start is only 4-byte aligned."""
        return t.detach().to(device=self.device, dtype=torch.float32, copy=True).contiguous()

    def _linear(self, w: torch.Tensor, b: Optional[torch.Tensor], stem: bool = False, mode: Optional[int] = None) -> Linear:
        """fp32 [N, K] -> 16-bit operand planes on the device (ser_split_bf16): bf16 hi (+ lo), or fp16 (hi + lo for FP16X)."""
        w = w.detach().to(torch.float32).contiguous()
        N, K = w.shape
        src = w.to(self.device)
        if mode is None:
            mode = self.stem_mode if stem else self.mode
        out = torch.empty((_PLANES[mode], N, K), dtype=_DTYPE[mode], device=self.device)
        if mode == _lib.MODE_FP16M:
            if K % 64:
                raise ValueError(f"FP16M weights need K % 64 == 0 (K = {K})")
            wscale = torch.zeros((K // 64, N), dtype=torch.int32, device=self.device)
            check(lib.ser_pack_f16m(src.data_ptr(), K, N, K, out.data_ptr(), K, N * K, wscale.data_ptr(), N, 1, None, _stream()), "ser_pack_f16m")
            torch.cuda.current_stream().synchronize()
            return Linear(out, None if b is None else self._dev_f32(b), N, K, wscale=wscale)
        check(lib.ser_split_bf16(src.data_ptr(), out.data_ptr(), N * K, mode, N * K, _stream()), "ser_split_bf16")
        torch.cuda.current_stream().synchronize()
        return Linear(out, None if b is None else self._dev_f32(b), N, K)

    def _linear_ln(self, w: torch.Tensor, b: Optional[torch.Tensor], ln_w: torch.Tensor, ln_b: torch.Tensor,
                   mode: Optional[int] = None) -> Linear:
        """Linear that consumes LayerNorm(x) given the RAW x (deferred LayerNorm, ser_hip.h):
        store W' = W * gamma, colsum(W') of exactly the bf16 planes the MFMAs will read, and
        t = beta W^T + b.  Load-time transform, like the weight-norm fold."""
        w64, g64, be64 = w.detach().double(), ln_w.detach().double(), ln_b.detach().double()
        lin = self._linear((w64 * g64[None, :]).float(), None, mode=mode)
        t = w64 @ be64
        if b is not None:
            t = t + b.detach().double()
        lin.b = self._dev_f32(t.float())
        if lin.wscale is not None:          # FP16M: the effective weight is w_hi + (the e4m3 image of) w_lo = the fp32 fold to ~2^-15
            lin.colsum = self._dev_f32((w64 * g64[None, :]).float().double().sum(dim=1).float())
        else:
            lin.colsum = lin.w.double().sum(dim=(0, 2)).float().contiguous()
        return lin

    def _new_act(self, rows, cols, zero=False, extra_rows=0, stem=False, mode: Optional[int] = None) -> Act:
        if mode is None:
            mode = self.stem_mode if stem else self.mode
        return Act(rows, cols, _PLANES[mode], self.device, zero=zero, extra_rows=extra_rows, dtype=_DTYPE[mode], mx=(mode == _lib.MODE_FP16M))

    # ------------------------------------------------------------------ launchers
    def _gemm(self, a: Act, lin: Linear, M: int, *, a_rowoff=None, lda=None, kc=0, ldj=0, groups=1,
              a_group_stride=0, w_group_stride=0, c_group_stride=0, N=None, K=None, act=_lib.ACT_NONE,
              residual=None, ldr=0, res_row_mod=0, out_f32=None, ldo_f32=0, out_act: Optional[Act] = None,
              out_rowmap=None, a_ptr_offset=0, k_algo=None, ln=None, ln_eps=1e-5, tile_cfg=0,
              ln_stats=None, ln_groups=0, stat_out=None, stat_groups=0, f32_col_begin=0,
              col_scale=1.0, col_scale_end=0, shift=None, ln_mean=None, stem=False, out_mode=0, mode=None, out_col=0,
              lnstat_out=None):
        rec = self._rec
        g = rec.slot("gemm") if rec is not None else GemmArgs()
        g.A = a.ptr + a_ptr_offset
        g.a_plane_stride = a.plane_stride
        g.a_rowoff = _ptr(a_rowoff)
        g.lda = a.cols if lda is None else lda
        g.kc, g.ldj = kc, ldj
        g.W = lin.w.data_ptr()
        g.w_plane_stride = lin.w.shape[1] * lin.w.shape[2]
        g.M, g.N, g.K = M, (lin.N if N is None else N), (lin.K if K is None else K)
        g.groups = groups
        g.a_group_stride, g.w_group_stride, g.c_group_stride = a_group_stride, w_group_stride, c_group_stride
        g.mode = mode if mode is not None else (self.stem_mode if stem else self.mode)
        g.out_mode = out_mode if out_mode != g.mode else 0
        g.bias = _ptr(lin.b)
        g.act = act
        g.residual = _ptr(residual)
        g.ldr = ldr
        g.res_row_mod = res_row_mod
        g.out_f32 = _ptr(out_f32)
        g.ldo_f32 = ldo_f32
        g.out_act = None if out_act is None else out_act.ptr + 2 * out_col     # out_col: first column written (16-bit elements)
        g.ldo_act = 0 if out_act is None else out_act.cols
        g.out_plane_stride = 0 if out_act is None else out_act.plane_stride
        g.out_rowmap = _ptr(out_rowmap)
        if ln is not None:                      # fused LayerNorm over the output row (conv stack)
            g.ln_gamma, g.ln_beta, g.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln_eps)
        g.tile_cfg = tile_cfg
        if ln_stats is not None:                # deferred LayerNorm of the A rows
            g.ln_stats_in, g.ln_groups, g.ln_colsum = ln_stats.data_ptr(), ln_groups, lin.colsum.data_ptr()
            g.ln_eps = float(self.geo.layer_norm_eps)
        if stat_out is not None:
            g.stat_out, g.stat_groups = stat_out.data_ptr(), stat_groups
        g.f32_col_begin = f32_col_begin
        g.col_scale, g.col_scale_end = float(col_scale), int(col_scale_end)
        if shift is not None:                   # producer side of the shifted operand copy (ser_hip.h): (mean of residual, shift_out, const)
            s_in, s_out, s_const = shift
            g.shift_in, g.shift_out, g.shift_const = _ptr(s_in), s_out.data_ptr(), float(s_const)
        if ln_mean is not None:                 # consumer side: (shift of the A rows, absolute row mean out)
            g.ln_shift, g.mean_out = _ptr(ln_mean[0]), ln_mean[1].data_ptr()
        g.lnstat_out = _ptr(lnstat_out)         # (relative mean, rstd) of the A rows for the attention kernel's in-kernel gate
        g.range_flag = self._flag if out_act is not None else None
        if g.mode == _lib.MODE_FP16M:           # block scales of both operands
            g.a_scale, g.a_scale_ld = a.scale.data_ptr(), a.scale_ld
            g.w_scale, g.w_scale_ld = lin.wscale.data_ptr(), lin.wscale.shape[1]
        if out_act is not None and (g.out_mode or g.mode) == _lib.MODE_FP16M:
            if out_col % 64:
                raise ValueError("an FP16M output must start on a 64-column tile")
            g.out_scale, g.out_scale_ld = out_act.scale.data_ptr() + 4 * (out_col // 64) * out_act.scale_ld, out_act.scale_ld
        if rec is not None:
            rec.commit(_lib.OP_GEMM, g, M=M)
            return
        if self.gemm_trace is None:
            check(lib.ser_gemm(C.byref(g), self._s()), "ser_gemm")
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ser_gemm(C.byref(g), self._s()), "ser_gemm")
        e1.record()
        # algorithmic FLOPs: 2*M*N*K over real (unpadded) channels, no tile-padding FLOPs
        k_real = g.K if k_algo is None else k_algo
        # algorithmic HBM bytes: every operand / result element touched exactly once
        planes = _PLANES[g.mode]
        nbytes = 2.0 * planes * (M * k_real * groups + g.N * groups * k_real)
        nbytes += 4.0 * M * g.N * groups * ((residual is not None) + (out_f32 is not None))
        nbytes += 2.0 * planes * M * g.N * groups * (out_act is not None)
        self.gemm_trace.append((e0, e1, 2.0 * M * g.N * groups * k_real, nbytes, _PRODUCTS[g.mode]))

    def _layernorm(self, x: torch.Tensor, ldx: int, ln, rows: int, D: int, *, gelu=False, out_f32=None,
                   out_act: Optional[Act] = None, eps=None, stem=False):
        g, b = ln
        mode = self.stem_mode if stem else self.mode
        if mode == _lib.MODE_FP16M:                   # ser_layernorm writes no FP16M copy; the encoders only need its fp32 output in that mode
            if out_act is not None:
                raise NotImplementedError("ser_layernorm has no FP16M operand copy")
            mode = _lib.MODE_FP16X
        eps = float(self.geo.layer_norm_eps if eps is None else eps)
        o_act = None if out_act is None else out_act.ptr
        ldo_act = 0 if out_act is None else out_act.cols
        ops = 0 if out_act is None else out_act.plane_stride
        ldo_f32 = D if out_f32 is not None else 0
        rec = self._rec
        a = rec.slot("layernorm") if rec is not None else _lib.LayerNormArgs()
        a.x, a.ldx, a.g, a.b, a.eps, a.gelu = x.data_ptr(), ldx, g.data_ptr(), b.data_ptr(), eps, int(gelu)
        a.out_f32, a.ldo_f32, a.out_act, a.ldo_act, a.out_plane_stride = _ptr(out_f32), ldo_f32, o_act, ldo_act, ops
        a.mode, a.rows, a.D = mode, rows, D
        a.range_flag = self._flag if out_act is not None else None
        if rec is not None:
            rec.commit(_lib.OP_LAYERNORM, a, rows=rows)
            return
        check(lib.ser_layernorm_v(C.byref(a), self._s()), "ser_layernorm")

    def _row_center(self, x: torch.Tensor, out_act: Act, stats: torch.Tensor, shift: torch.Tensor, rows: int, D: int):
        """hidden_states[0] -> centred operand copy + row partials + shift for encoder layer 0 (ser_row_center)."""
        groups = stats.shape[1]
        rec = self._rec
        if rec is not None:
            a = rec.slot("row_center")
            a.x, a.ldx, a.out_act, a.ldo_act, a.out_plane_stride = x.data_ptr(), D, out_act.ptr, out_act.cols, out_act.plane_stride
            a.stats, a.shift, a.stat_groups, a.mode, a.rows, a.D = stats.data_ptr(), shift.data_ptr(), groups, self._lay_modes(0)["x_mode"], rows, D
            if out_act.scale is not None:
                a.out_scale, a.out_scale_ld = out_act.scale.data_ptr(), out_act.scale_ld
            a.range_flag = self._flag
            rec.commit(_lib.OP_ROW_CENTER, a, rows=rows)
            return
        a = _lib.RowCenterArgs()
        a.x, a.ldx, a.out_act, a.ldo_act, a.out_plane_stride = x.data_ptr(), D, out_act.ptr, out_act.cols, out_act.plane_stride
        a.stats, a.shift, a.stat_groups, a.mode, a.rows, a.D = stats.data_ptr(), shift.data_ptr(), groups, self._lay_modes(0)["x_mode"], rows, D
        if out_act.scale is not None:
            a.out_scale, a.out_scale_ld = out_act.scale.data_ptr(), out_act.scale_ld
        a.range_flag = self._flag
        check(lib.ser_row_center_v(C.byref(a), self._s()), "ser_row_center")

    def _attention(self, qkv: Act, frame_offs_dev, B, max_frames, out: Act, *, table=None, table_T=0, gate=None,
                   gru_const=None, key_lens=None, bias2d=None, gate_in=None, out_m=False):
        """``gate_in`` = (operand copy of the layer input, lnstat of the packed projection, folded weights, constants): the WavLM gate
        is computed inside the kernel (ser_attention_args.gate_x) instead of read from gate columns of ``qkv``."""
        D, H, dh = self.geo.hidden, self.geo.heads, self.geo.head_dim
        # column blocks of the packed projection: [q | k | v | gate] or, with the logit path on its own launch ("f16q"), [q | k | gate | v]
        q_col, k_col, v_col, gate_col = self._qkv_cols()
        amode = _lib.MODE_FP16Q if self.qk_mode is not None else self.attn_mode
        rec = self._rec
        a = rec.slot("attention") if rec is not None else _lib.AttentionArgs()
        a.qkv, a.ld, a.plane_stride = qkv.ptr, qkv.cols, qkv.plane_stride
        a.q_col, a.k_col, a.v_col, a.B = q_col, k_col, v_col, B
        a.frame_offs, a.table, a.gate = frame_offs_dev.data_ptr(), _ptr(table), _ptr(gate)
        a.max_frames, a.table_T = max_frames, table_T
        a.out, a.ldo, a.out_plane_stride = out.ptr, out.cols, out.plane_stride
        a.H, a.dh, a.scale, a.mode, a.gate_col = H, dh, -1.0, amode, gate_col       # q is pre-scaled
        if out_m:                                                                   # context rows as SER_MODE_FP16M operands (ABI 14)
            a.out_mode, a.out_scale, a.out_scale_ld = _lib.MODE_FP16M, out.scale.data_ptr(), out.scale_ld
        else:
            a.out_mode, a.out_scale, a.out_scale_ld = 0, None, 0
        a.gru_const, a.key_lens = _ptr(gru_const), _ptr(key_lens)
        a.bias2d, a.bias2d_ld = _ptr(bias2d), (0 if bias2d is None else bias2d.shape[-1])
        if gate_in is not None:
            xa, lnstat, gw, gcb = gate_in
            a.gate_x, a.gate_x_ld, a.gate_x_plane_stride, a.gate_x_planes = xa.ptr, xa.cols, xa.plane_stride, xa.planes
            a.gate_stat, a.gate_w, a.gate_cb = lnstat.data_ptr(), gw.data_ptr(), gcb.data_ptr()
            a.gate_w_plane_stride = gw.shape[1] * gw.shape[2]
        if rec is not None:
            rec.commit(_lib.OP_ATTENTION, a, B=B, max_frames=max_frames, table_T=table_T)
            return
        check(lib.ser_attention_v(C.byref(a), self._s()), "ser_attention")

    @staticmethod
    def _gate_in(pl, lay):
        return (pl["xa"], pl["gst"], lay["gate_w"], lay["gate_cb"]) if "gate_w" in lay else None

    def _gate_pad(self) -> int:
        """extra columns of the packed projection: the WavLM gate's two pre-activations per head, padded to a multiple of 8"""
        in_cols = any(not self._lay_modes(i)["gate_in_attn"] for i in range(self.geo.num_layers))   # some layer keeps the gate in the projection
        return ((2 * self.geo.heads + 7) // 8) * 8 if (self.geo.family == FAMILY_WAVLM and in_cols) else 0

    def _qkv_cols(self):
        """(q, k, v, gate) first columns inside the packed projection output"""
        D = self.geo.hidden
        if self.qk_mode is None:
            return 0, D, 2 * D, 3 * D
        return 0, D, 2 * D + self._gate_pad(), 2 * D

    def _qkv_gemm(self, pl, lay, M: int, first: bool, gx: int, ln_mean) -> None:
        """packed Q K V (+ gate) projection with LayerNorm 1 deferred into it; q leaves multiplied by dh^-0.5 * log2(e).
        "f16q": two launches over the same operand copy -- [q | k | gate] as the 3-product FP16X GEMM on both planes of x,
        [v] as a single-product FP16 GEMM on its hi plane."""
        geo = self.geo
        D = geo.hidden
        stats = pl["px0"] if first else pl["px"]
        scale = geo.head_dim ** -0.5 * 1.4426950408889634
        lnstat = pl["gst"] if "gate_w" in lay else None     # (relative mean, rstd) per row for the attention kernel's in-kernel gate
        if self.qk_mode is None:
            self._gemm(pl["xa"], lay["qkv"], M, ln_stats=stats, ln_groups=gx, out_act=pl["qkv"], col_scale=scale,
                       col_scale_end=D, ln_mean=ln_mean, mode=lay["qkv_mode"], out_mode=lay["qkv_out_mode"], lnstat_out=lnstat)
            return
        self._gemm(pl["xa"], lay["qk"], M, ln_stats=stats, ln_groups=gx, out_act=pl["qkv"], col_scale=scale,
                   col_scale_end=D, ln_mean=ln_mean, mode=self.qk_mode, lnstat_out=lnstat)
        self._gemm(pl["xa"], lay["v"], M, ln_stats=stats, ln_groups=gx, out_act=pl["qkv"], out_col=self._qkv_cols()[2])

    @staticmethod
    def _stat_groups(n_cols: int, groups: int = 1) -> int:
        """64-column partial-sum slots a GEMM output of `groups` x `n_cols` columns produces (made even)."""
        g = groups * ((n_cols + 63) // 64)
        return g + (g & 1)

    def _run_layers(self, pl, states, first_groups: int, B: int, max_frames: int, last_state: Optional[int] = None):
        """Pre-LN / stable-LN encoder layers with BOTH LayerNorms deferred into the consuming GEMMs:
        x -> [QKV(+gate) GEMM: LN1 folded] -> attention -> [out GEMM +x -> h] -> [FC1 GEMM: LN2 folded, GELU]
          -> [FC2 GEMM +h -> next x].  Producers emit the bf16 operand copy and the row partial sums, both SHIFTED by
        the row mean of their residual input: the consumer of x (QKV) reports x's absolute row mean (mx), the producer
        of h (out-proj) shifts by it and records the shift (sh), the consumer of h (FC1) reports mh, the producer of
        the next x (FC2) shifts by that (sx).  Offsets that live in the residual stream never reach the bf16 rounding or
        the one-pass variance.  states[0] is centred once by ser_row_center."""
        geo = self.geo
        M, D, L = pl["M"], geo.hidden, geo.num_layers
        wavlm = geo.family == FAMILY_WAVLM
        gD = self._stat_groups(D)
        gx = first_groups
        # ``last_state`` = N: the caller reads hidden_states[N] only (the reference's speech script keeps one state,
        # preprocess_speech.py:67), so the layers that only feed later states are not launched.  Recording a command list marks
        # where every state is complete instead (Tape.marks), and the replay stops there.
        rec = self._rec
        if rec is not None:
            rec.marks[0] = rec.n
        elif last_state == 0:
            return
        self._row_center(states[0], pl["xa"], pl["px0"], pl["sx"], M, D)
        shifted = not _NO_SHIFT
        for i, lay in enumerate(self.layers):
            x = states[i]
            last = i + 1 == L
            nxt = pl["last"] if last else states[i + 1]
            if self.block_trace is not None:
                b0 = torch.cuda.Event(enable_timing=True)
                b0.record()
            self._qkv_gemm(pl, lay, M, i == 0, gx, (pl["sx"], pl["mx"]) if shifted else None)
            if wavlm:
                self._attention(pl["qkv"], pl["frame_offs"], B, max_frames, pl["ctx"], table=pl["table"],
                                table_T=pl["Tmax"], gru_const=lay["gate_c"], gate_in=self._gate_in(pl, lay), out_m=lay["out_m"])
            else:
                self._attention(pl["qkv"], pl["frame_offs"], B, max_frames, pl["ctx"], out_m=lay["out_m"])
            self._gemm(pl["ctx"], lay["out"], M, residual=x, ldr=D, out_f32=pl["h"], ldo_f32=D,
                       out_act=pl["ha"], stat_out=pl["ph"], stat_groups=gD, mode=_lib.MODE_FP16M if lay["out_m"] else self.attn_mode, out_mode=self.mode,
                       shift=(pl["mx"], pl["sh"], lay["out_bias_mean"]) if shifted else None)
            if self.block_trace is not None:
                b1 = torch.cuda.Event(enable_timing=True)
                b1.record()
                self.block_trace.append((b0, b1, B))
            self._gemm(pl["ha"], lay["fc1"], M, ln_stats=pl["ph"], ln_groups=gD, act=_lib.ACT_GELU, out_act=pl["ffn"],
                       ln_mean=(pl["sh"], pl["mh"]) if shifted else None)
            if last:
                self._gemm(pl["ffn"], lay["fc2"], M, residual=pl["h"], ldr=D, out_f32=nxt, ldo_f32=D)
            else:
                self._gemm(pl["ffn"], lay["fc2"], M, residual=pl["h"], ldr=D, out_f32=nxt, ldo_f32=D,
                           out_act=pl["xa"], stat_out=pl["px"], stat_groups=gD, out_mode=self.layers[i + 1]["x_mode"],
                           shift=(pl["mh"], pl["sx"], lay["fc2_bias_mean"]) if shifted else None)
            gx = gD
            if not last:
                if rec is not None:
                    rec.marks[i + 1] = rec.n
                elif last_state == i + 1:
                    return
        self._layernorm(pl["last"], D, self.enc_ln, M, D, out_f32=states[L])

    def recorded_tape(self, lengths: Sequence[int], slot: int) -> Tape:
        """the command list the slot's last forward over ``lengths`` replayed (SpeechEncoder: per arena; Whisper: per plan)"""
        pl = self._plan(lengths, slot)
        tape = pl.get("tape") or getattr(self, "_arenas", {}).get(slot, {}).get("tape")
        if tape is None:
            raise _lib.SerHipError("no recorded command list for this slot yet (run a forward first)")
        return tape

    @_on_stream
    def attention_blocks_only(self, lengths: Sequence[int], slot: int) -> int:
        """Measurement aid (bench.py ``attention_block``): launch ONLY the attention sub-graph of every layer -- packed QKV(+gate)
        projection -> attention -> output projection -- over the buffers the slot's last forward left behind (same shapes,
        same kernels, stale but finite data), on the current stream.  Returns the number of layer calls."""
        pl = self._plan(lengths, slot)
        self._guard_word(pl)
        geo = self.geo
        M, D, B = pl["M"], geo.hidden, len(lengths)
        max_frames = pl.get("Tmax", geo.max_source_positions)
        wavlm = geo.family == FAMILY_WAVLM
        gD = self._stat_groups(D)
        for i, lay in enumerate(self.layers):
            self._qkv_gemm(pl, lay, M, i == 0, pl["first_groups"] if i == 0 else gD, (pl["sx"], pl["mx"]))
            if wavlm:
                self._attention(pl["qkv"], pl["frame_offs"], B, max_frames, pl["ctx"], table=pl["table"], table_T=pl["Tmax"],
                                gru_const=lay["gate_c"], gate_in=self._gate_in(pl, lay), out_m=lay["out_m"])
            else:
                self._attention(pl["qkv"], pl["frame_offs"], B, max_frames, pl["ctx"], out_m=lay["out_m"])
            self._gemm(pl["ctx"], lay["out"], M, residual=pl["h"], ldr=D, out_f32=pl["h"], ldo_f32=D, out_act=pl["ha"],
                       stat_out=pl["ph"], stat_groups=gD, shift=(pl["mx"], pl["sh"], lay["out_bias_mean"]),
                       mode=_lib.MODE_FP16M if lay["out_m"] else self.attn_mode, out_mode=self.mode)
        return len(self.layers)

    def _layer_weights(self, sd, p: str, a: str, ln1: str, ln2: str, fc1: str, fc2: str, k_bias: bool, gate: bool, index: int = 0):
        """One encoder layer's GEMM operands; LN1 folds into the packed QKV (+gate) projection, LN2 into FC1."""
        D, H, dh = self.geo.hidden, self.geo.heads, self.geo.head_dim
        lm = self._lay_modes(index)
        wdev = sd[a + ".q_proj.weight"].device            # CPU state dict, or device views of the broadcast bucket (dist.py)
        kb = sd[a + ".k_proj.bias"] if k_bias else torch.zeros(D, device=wdev)
        ws = [sd[a + ".q_proj.weight"], sd[a + ".k_proj.weight"], sd[a + ".v_proj.weight"]]
        bs = [sd[a + ".q_proj.bias"], kb, sd[a + ".v_proj.bias"]]
        lay = dict(qkv_mode=lm["qkv_mode"], x_mode=lm["x_mode"], qkv_out_mode=lm["qkv_out_mode"], out_m=lm["out_m"])
        if gate and lm["gate_in_attn"]:
            # WavLM GRU gate (HF modeling_wavlm.py:167-180): its two pre-activations per head are linear in LN1(x) restricted to the
            # head's dh channels.  ser_attention evaluates them per query from the layer input's operand copy with the LayerNorm in
            # closed form (ser_attention_args.gate_x): pre_j = rstd (x . (gamma w_j) - mean sum(gamma w_j)) + (beta . w_j + b_j)
            from .weights import fold_wavlm_gate
            wg, t = fold_wavlm_gate(sd[a + ".gru_rel_pos_linear.weight"], sd[a + ".gru_rel_pos_linear.bias"],
                                    sd[ln1 + ".weight"], sd[ln1 + ".bias"], H, dh)
            # operand planes [planes][H][2][dh] in the format the attention launch multiplies in (one MFMA chain per query block)
            gmode = self.qk_mode if self.qk_mode is not None else self.attn_mode
            glin = self._linear(wg.float(), None, mode=gmode)
            cs = glin.w.double().sum(dim=(0, 2)).view(H, 2)                             # column sums of exactly the planes the MFMAs read
            lay["gate_w"] = glin.w
            lay["gate_cb"] = self._dev_f32(torch.cat([cs.to(t.device), t], 1).float())
            lay["gate_c"] = self._dev_f32(sd[a + ".gru_rel_pos_const"].reshape(-1))
            gate = False                                                                # no gate columns in the packed projection
        if gate:
            # (SER_GATE_IN_ATTN=0) the two pre-activations per head as 2H extra output columns of the packed projection
            w8, b8 = sd[a + ".gru_rel_pos_linear.weight"].float(), sd[a + ".gru_rel_pos_linear.bias"].float()
            wa, wb = w8[:4].sum(0), w8[4:].sum(0)
            wg = torch.zeros(2 * H, D, device=wdev)
            for h in range(H):
                wg[2 * h, h * dh:(h + 1) * dh] = wa
                wg[2 * h + 1, h * dh:(h + 1) * dh] = wb
            pad = (-2 * H) % 8                                   # GEMM N must stay a multiple of 8
            ws.append(torch.cat([wg, torch.zeros(pad, D, device=wdev)], 0))
            bs.append(torch.cat([torch.stack([b8[:4].sum(), b8[4:].sum()]).repeat(H), torch.zeros(pad, device=wdev)]))
            lay["gate_c"] = self._dev_f32(sd[a + ".gru_rel_pos_const"].reshape(-1))
        if self.qk_mode is None:
            lay["qkv"] = self._linear_ln(torch.cat(ws, 0), torch.cat(bs, 0), sd[ln1 + ".weight"], sd[ln1 + ".bias"], mode=lm["qkv_mode"])
        else:
            # logit path [q | k | gate] on fp16 hi + lo planes (3 products), [v] on one fp16 plane
            lay["qk"] = self._linear_ln(torch.cat(ws[:2] + ws[3:], 0), torch.cat(bs[:2] + bs[3:], 0), sd[ln1 + ".weight"],
                                        sd[ln1 + ".bias"], mode=self.qk_mode)
            lay["v"] = self._linear_ln(ws[2], bs[2], sd[ln1 + ".weight"], sd[ln1 + ".bias"])
        lay["out"] = self._linear(sd[a + ".out_proj.weight"], sd[a + ".out_proj.bias"], mode=_lib.MODE_FP16M if lm["out_m"] else self.attn_mode)
        lay["fc1"] = self._linear_ln(sd[fc1 + ".weight"], sd[fc1 + ".bias"], sd[ln2 + ".weight"], sd[ln2 + ".bias"])
        lay["fc2"] = self._linear(sd[fc2 + ".weight"], sd[fc2 + ".bias"])
        # load-time part of the operand shift: a uniform offset in a bias moves the row mean by exactly its mean
        lay["out_bias_mean"] = float(sd[a + ".out_proj.bias"].double().mean())
        lay["fc2_bias_mean"] = float(sd[fc2 + ".bias"].double().mean())
        return lay

    def _layer_buffers(self, pl, M: int, first_groups: int):
        geo, dev = self.geo, self.device
        D, Fd = geo.hidden, geo.ffn
        nqkv = 3 * D + self._gate_pad()
        gD = self._stat_groups(D)
        # (an FP16M buffer serves FP16X layers too: same two fp16-sized planes, the block-scale words are simply not touched)
        x_modes = {self._lay_modes(i)["x_mode"] for i in range(geo.num_layers)}
        pl["xa"] = self._new_act(M, D, mode=_lib.MODE_FP16M if _lib.MODE_FP16M in x_modes else self.x_mode)
        pl["ha"] = self._new_act(M, D)
        if self.fp16_planes:
            pl["range_flag"] = torch.zeros(1, dtype=torch.int32, device=dev)   # the slot's fp16 range-guard word (HiddenStates.range_flag)
        # row partial sums (sum, sum^2 per 64-column group).  One buffer per producer layout: a padding
        # slot (odd group count) is never written and must stay zero.
        pl["px0"] = torch.zeros((M, first_groups, 2), dtype=torch.float32, device=dev)   # states[0] (ser_row_center)
        pl["sx"] = torch.zeros(M, dtype=torch.float32, device=dev)       # row shift of the xa copy / px partials
        pl["sh"] = torch.zeros(M, dtype=torch.float32, device=dev)       # row shift of the ha copy / ph partials
        pl["mx"] = torch.zeros(M, dtype=torch.float32, device=dev)       # absolute row mean of x (written by the QKV GEMM)
        pl["mh"] = torch.zeros(M, dtype=torch.float32, device=dev)       # absolute row mean of h (written by the FC1 GEMM)
        pl["gst"] = torch.zeros((M, 2), dtype=torch.float32, device=dev)  # (relative mean, rstd) of x's rows (packed projection -> attention's gate)
        pl["px"] = torch.zeros((M, gD, 2), dtype=torch.float32, device=dev)              # FC2 outputs
        pl["ph"] = torch.zeros((M, gD, 2), dtype=torch.float32, device=dev)              # out-proj outputs
        pl["qkv"] = self._new_act(M, nqkv, mode=self._lay_modes(geo.num_layers - 1)["qkv_out_mode"])   # "f16q": q, k, gate columns carry a lo plane, v's stays unused
        any_out_m = any(self._lay_modes(i)["out_m"] for i in range(geo.num_layers))
        pl["ctx"] = self._new_act(M, D, mode=_lib.MODE_FP16M if any_out_m else self.attn_mode)   # (an FP16M buffer serves the FP16X layers too)
        pl["h"] = torch.empty((M, D), dtype=torch.float32, device=dev)
        pl["ffn"] = self._new_act(M, Fd)
        pl["last"] = torch.empty((M, D), dtype=torch.float32, device=dev)

    def capture(self, packed_wave: torch.Tensor, lengths: Sequence[int]):
        """Record one forward over a fixed batch shape into a hipGraph (via torch's CUDAGraph
        plumbing).  A forward is ~230 short launches; replaying the graph removes the host-side
        launch gaps that otherwise cost ~25 % of a step.  Returns (graph, hidden_states): the
        states tensor is overwritten in place by every ``graph.replay()``."""
        self._plan(lengths)                                  # buffers + offset tables allocated before capture
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.forward(packed_wave, lengths)               # warm-up on the side stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            hs = self.forward(packed_wave, lengths)
        return graph, hs

    def capture_concurrent(self, micro_batches):
        """Record the forwards of several independent micro-batches as PARALLEL branches of one
        hipGraph (one HIP stream each).  Every GEMM of this path spends a third of its time in a
        memory-bound prologue/epilogue during which the matrix cores idle; with two utterance
        groups in flight the tail of one group's kernel overlaps the main loop of the other's.
        ``micro_batches`` = [(packed_wave, lengths), ...]; returns (graph, [HiddenStates, ...])."""
        for slot, (wave, lengths) in enumerate(micro_batches):
            self._plan(lengths, slot)
        warm = torch.cuda.Stream(device=self.device)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for slot, (wave, lengths) in enumerate(micro_batches):
                self.forward(wave, lengths, slot=slot)
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        sides = [torch.cuda.Stream(device=self.device) for _ in micro_batches[1:]]
        graph = torch.cuda.CUDAGraph()
        outs = [None] * len(micro_batches)
        with torch.cuda.graph(graph):
            main = torch.cuda.current_stream()
            for st in sides:
                st.wait_stream(main)                                   # fork
            for slot, st in enumerate(sides, start=1):
                with torch.cuda.stream(st):
                    outs[slot] = self.forward(*micro_batches[slot], slot=slot)
            outs[0] = self.forward(*micro_batches[0], slot=0)
            for st in sides:
                main.wait_stream(st)                                   # join
        return graph, outs

    def _ln_pair(self, sd, prefix):
        return (self._dev_f32(sd[prefix + ".weight"]), self._dev_f32(sd[prefix + ".bias"]))


def _fold_weight_norm(sd) -> torch.Tensor:
    """w = g * v / ||v||_(dims 0,1) per tap (HF modeling_wavlm.py:48-75), at load time."""
    base = "encoder.pos_conv_embed.conv."
    if base + "parametrizations.weight.original0" in sd:
        g, v = sd[base + "parametrizations.weight.original0"], sd[base + "parametrizations.weight.original1"]
    elif base + "weight_g" in sd:
        g, v = sd[base + "weight_g"], sd[base + "weight_v"]
    else:
        return sd[base + "weight"].float()
    g, v = g.float(), v.float()
    return v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())


class SpeechEncoder(_EncoderBase):
    """WavLM / wav2vec2 / HuBERT (stable-LayerNorm, layer-norm conv stack) on libserhip."""

    def __init__(self, geo: EncoderGeometry, state_dict, device="cuda:0", mode: str = "bf16"):
        super().__init__(geo, device, mode)
        if geo.family == FAMILY_WHISPER:
            raise ValueError("use WhisperEncoder for the whisper family")
        if not geo.feat_proj_layer_norm:
            raise NotImplementedError("feat_proj_layer_norm=False checkpoints are not supported")
        sd = state_dict
        D, C0 = geo.hidden, geo.conv_dim[0]
        if any(c != C0 for c in geo.conv_dim):
            raise NotImplementedError("conv_dim must be uniform")
        # conv stack
        p0 = "feature_extractor.conv_layers.0"
        self.conv0_w = self._dev_f32(sd[p0 + ".conv.weight"].reshape(C0, geo.conv_kernel[0]))
        self.conv0_b = self._dev_f32(sd[p0 + ".conv.bias"]) if geo.conv_bias else None
        # matrix-core form of conv layer 0: taps zero-padded to K = 64 (frames come from ser_wave_frames)
        if geo.conv_kernel[0] > 64:
            raise NotImplementedError("conv layer 0 kernel wider than 64 taps")
        w0 = torch.zeros((C0, 64), dtype=torch.float32)
        w0[:, : geo.conv_kernel[0]] = sd[p0 + ".conv.weight"].reshape(C0, geo.conv_kernel[0]).float()
        self.conv0 = self._linear(w0, sd[p0 + ".conv.bias"] if geo.conv_bias else None, stem=True)
        self.conv_ln = [self._ln_pair(sd, f"feature_extractor.conv_layers.{i}.layer_norm") for i in range(len(geo.conv_dim))]
        self.convs: List[Linear] = []
        for i in range(1, len(geo.conv_dim)):
            p = f"feature_extractor.conv_layers.{i}.conv"
            w = sd[p + ".weight"].float()                                  # [Cout, Cin, k]
            w2 = w.permute(0, 2, 1).reshape(w.shape[0], -1)                # K index = tap*Cin + c
            self.convs.append(self._linear(w2, sd[p + ".bias"] if geo.conv_bias else None, stem=True))
        self.proj_ln = self._ln_pair(sd, "feature_projection.layer_norm")
        self.proj = self._linear(sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"], stem=True)
        # positional conv: per group [Cg out][tap][Cg in padded to a multiple of 64]
        G, k = geo.pos_conv_groups, geo.pos_conv_kernel
        Cg = D // G
        self.pos_cg, self.pos_kc = Cg, ((Cg + 63) // 64) * 64
        w = _fold_weight_norm(sd)                                          # [D, Cg, k]
        wp = torch.zeros((G, Cg, k, self.pos_kc), dtype=torch.float32)
        wp[:, :, :, :Cg] = w.view(G, Cg, Cg, k).permute(0, 1, 3, 2)
        # The positional conv runs in the LAYER format when its dot products are short enough: in "f16" mode it then costs 3x
        # less than on the fp32x stem and, unlike the conv stack and the projection, moves the error by nothing measurable
        # (WavLM-large, K = 128 taps x 64 channels: 6.8e-4 either way; HuBERT-xlarge, 128 x 80: 8.0e-4 either way).  XLS-R-2B's
        # 128 x 120 = 15 360-long sums do feel fp16 operands (7.0e-4 -> 8.3e-4 at full geometry), so they stay on the stem format.
        # "f16q" / "f16a" keep it on the stem format always: an error in hidden_states[0] enters layer 0's logits, and these modes
        # exist for attention maps sharp enough to amplify it (LoRA stress fixture: 1.4e-3 -> see DESIGN.md section 4)
        self.pos_in_stem = (self.mode_name == "f16" and Cg * k > 128 * 80) or self.mode_name in ("f16q", "f16a", "f16m", "f16mf")
        self.pos = self._linear(wp.reshape(G * Cg, k * self.pos_kc), sd["encoder.pos_conv_embed.conv.bias"], stem=self.pos_in_stem)
        self.enc_ln = self._ln_pair(sd, "encoder.layer_norm")
        self.layers = []
        for i in range(geo.num_layers):
            p = f"encoder.layers.{i}"
            self.layers.append(self._layer_weights(
                sd, p, p + ".attention", p + ".layer_norm", p + ".final_layer_norm",
                p + ".feed_forward.intermediate_dense", p + ".feed_forward.output_dense",
                k_bias=True, gate=(geo.family == FAMILY_WAVLM), index=i))
        if geo.family == FAMILY_WAVLM:
            self.rel_embed = self._dev_f32(sd["encoder.layers.0.attention.rel_attn_embed.weight"])

    # ------------------------------------------------------------------ batch plan
    # ------------------------------------------------------------------ per-slot arenas + plans
    def _arena(self, slot: int, need: Dict[str, int]):
        """Grow-only buffer set of one slot, sized by the largest batch seen so far: a stream of ragged batches
        re-uses the same HBM instead of allocating ~25 tensors per batch.  ``need``: rows after conv 0 / conv 1,
        frames M, halo'd rows, utterances B, longest utterance Tmax."""
        arenas = self.__dict__.setdefault("_arenas", {})
        ar = arenas.get(slot)
        if ar is not None and all(ar["cap"][k] >= v for k, v in need.items()):
            return ar
        cap = {k: max(int(v), ar["cap"][k] if ar else 0) for k, v in need.items()}
        geo, dev = self.geo, self.device
        C0, D = geo.conv_dim[0], geo.hidden
        nl = len(geo.conv_dim)
        ar = dict(cap=cap)
        ar["frames"] = self._new_act(cap["rows0"], 64, stem=True)
        ar["wave_work"] = torch.empty(lib.ser_workspace_bytes(_lib.WS_WAVE_FRAMES, cap["B"], 0, 0, 0, self.stem_mode),
                                      dtype=torch.uint8, device=dev)
        ar["conv_act"] = [self._new_act(cap["rows0"], C0, stem=True), self._new_act(cap["rows1"], C0, stem=True)]       # ping-pong
        ar["conv_rowoff"] = [torch.empty(cap["rows1"], dtype=torch.int32, device=dev) for _ in range(1, nl)]
        ar["halo_rowmap"] = torch.empty(cap["M"], dtype=torch.int32, device=dev)
        ar["pos_rowoff"] = torch.empty(cap["M"], dtype=torch.int32, device=dev)
        ar["feat_f32"] = torch.empty((cap["M"], C0), dtype=torch.float32, device=dev)
        ar["feat_act"] = self._new_act(cap["M"], C0, stem=True)
        ar["proj_f32"] = torch.empty((cap["M"], D), dtype=torch.float32, device=dev)
        ar["halo_act"] = self._new_act(cap["halo"], D, zero=True, extra_rows=1, stem=self.pos_in_stem)
        ar["states"] = torch.empty((geo.num_layers + 1, cap["M"], D), dtype=torch.float32, device=dev)
        ar["first_groups"] = 2                               # ser_row_center writes one (sum, sum^2) slot + one zero slot
        self._layer_buffers(ar, cap["M"], ar["first_groups"])
        if geo.family == FAMILY_WAVLM:
            ar["table"] = torch.empty(geo.heads * (2 * cap["Tmax"] - 1), dtype=torch.float32, device=dev)
        # small per-batch tables: one pinned host blob -> one async H2D into one device blob (int64 words)
        # fixed regions of cap B + 2 words each (pointers into the blob must not move from batch to batch):
        # sample_offs | offs[0..nl) (int32, two per word) | conv bases [1..nl) | halo base | pos base
        ar["tab_region"] = cap["B"] + 2
        words = (1 + nl + (nl - 1) + 2) * ar["tab_region"]
        ar["tab_host"] = torch.empty(words, dtype=torch.int64).pin_memory()
        ar["tab_dev"] = torch.empty(words, dtype=torch.int64, device=dev)
        ar["tab_evt"] = None
        ar["plan"] = None                                   # (lengths, plan) currently laid out in this arena
        arenas[slot] = ar
        return ar

    def _plan(self, lengths: Sequence[int], slot: int = 0):
        """Row bookkeeping of one ragged batch in slot ``slot``'s arena.  Only O(B) integers are computed on the
        host and uploaded (one pinned blob, one async copy); the O(rows) row tables are built on the device by
        ``ser_ragged_index``.  A repeat of the previous lengths (benchmark, graph replay) re-uses the laid-out plan.
        The returned buffers -- including the hidden states -- stay valid until the next plan of the same slot."""
        lengths = tuple(int(n) for n in lengths)
        geo, dev = self.geo, self.device
        B = len(lengths)
        nl = len(geo.conv_dim)
        chains = [geo.frame_chain(n) for n in lengths]
        if B == 0 or min(c[-1] for c in chains) < 1:
            raise ValueError("utterance shorter than the conv stack's receptive field (400 samples)")
        offs = [np.concatenate([[0], np.cumsum([c[i] for c in chains])]).astype(np.int64) for i in range(nl)]
        T = [c[-1] for c in chains]
        M, Tmax = int(offs[-1][-1]), max(T)
        half = geo.pos_conv_kernel // 2
        halo_rows = int(M + half * (B + 1))
        ar = self._arena(slot, dict(rows0=int(offs[0][-1]), rows1=int(offs[1][-1]) if nl > 1 else 1, M=M,
                                    halo=halo_rows, B=B, Tmax=Tmax))
        if ar["plan"] is not None and ar["plan"][0] == lengths:
            return ar["plan"][1]
        st = _stream()
        C0, D = geo.conv_dim[0], geo.hidden
        # ---- O(B) tables -> pinned blob (int64 words; int32 tables packed two per word)
        if ar["tab_evt"] is not None:
            ar["tab_evt"].synchronize()                     # the previous batch's copy out of the blob has finished
        host64 = ar["tab_host"].numpy()
        host32 = host64.view(np.int32)
        dev64 = ar["tab_dev"]
        dev32 = dev64.view(torch.int32)
        region = [0]                                        # next fixed region (int64 words)

        def put64(a):
            a = np.asarray(a, dtype=np.int64)
            w = region[0] * ar["tab_region"]
            region[0] += 1
            host64[w:w + len(a)] = a
            return dev64[w:w + len(a)]

        def put32(a):
            a = np.asarray(a, dtype=np.int32)
            w = region[0] * ar["tab_region"]
            region[0] += 1
            host32[2 * w:2 * w + len(a)] = a
            return dev32[2 * w:2 * w + len(a)]

        pl = dict(ar)                                       # arena buffers + this batch's views / numbers
        pl["rows"] = [Sz(int(o[-1]), f"rows{i}") for i, o in enumerate(offs)]
        pl["sizes"] = {f"rows{i}": int(r) for i, r in enumerate(pl["rows"])}
        pl["sizes"].update(M=M, B=B, Tmax=Tmax)
        pl.update(B=Sz(B, "B"), lengths=lengths, T=T, M=Sz(M, "M"), Tmax=Sz(Tmax, "Tmax"), halo_rows=halo_rows)
        pl["sample_offs"] = put64(np.concatenate([[0], np.cumsum(lengths)]))
        offs_dev = [put32(o) for o in offs]                 # offs[i][b] = first row of utterance b after conv layer i
        pl["frame_offs0"] = offs_dev[0]
        pl["frame_offs"] = offs_dev[-1]
        pl["frame_offs_host"] = [int(x) for x in offs[-1]]
        starts = np.array([half * (b + 1) + offs[-1][b] for b in range(B)], dtype=np.int64)
        base_conv = [put64(offs[i - 1][:B]) for i in range(1, nl)]
        base_halo, base_pos = put64(starts), put64(starts - half)
        dev64.copy_(ar["tab_host"], non_blocking=True)
        ar["tab_evt"] = torch.cuda.Event()
        ar["tab_evt"].record()
        # ---- O(rows) tables on the device
        rowoffs = []
        for i in range(1, nl):
            out = ar["conv_rowoff"][i - 1][: pl["rows"][i]]
            check(lib.ser_ragged_index(offs_dev[i].data_ptr(), base_conv[i - 1].data_ptr(), B, geo.conv_stride[i], C0, 8,
                                       out.data_ptr(), pl["rows"][i], st), "ser_ragged_index")
            rowoffs.append(out)
        pl["conv_rowoff"] = rowoffs
        # halo'd layout of the positional-conv input: [64 zero rows][utt 0][64 zero rows][utt 1] ... [64 zero rows]
        check(lib.ser_ragged_index(offs_dev[-1].data_ptr(), base_halo.data_ptr(), B, 1, 1, 1,
                                   ar["halo_rowmap"].data_ptr(), M, st), "ser_ragged_index")
        check(lib.ser_ragged_index(offs_dev[-1].data_ptr(), base_pos.data_ptr(), B, 1, D, 8,
                                   ar["pos_rowoff"].data_ptr(), M, st), "ser_ragged_index")
        if ar["plan"] is not None:
            ar["halo_act"].t.zero_()                        # the halo rows sit elsewhere than in the previous batch
        pl["states"] = ar["states"][:, :M]
        if geo.family == FAMILY_WAVLM:
            pl["table"] = ar["table"][: geo.heads * (2 * Tmax - 1)].view(geo.heads, 2 * Tmax - 1)
            if ar.get("table_T") != Tmax:
                check(lib.ser_wavlm_bias_table(self.rel_embed.data_ptr(), pl["table"].data_ptr(), Tmax, geo.heads,
                                               geo.num_buckets, geo.max_bucket_distance, st), "ser_wavlm_bias_table")
                ar["table_T"] = Tmax
        ar["plan"] = (lengths, pl)
        return pl

    # ------------------------------------------------------------------ forward
    def upload(self, waves: Sequence[np.ndarray], slot: int = 0) -> torch.Tensor:
        """Pack raw fp32 waveforms into slot ``slot``'s (reused, grow-only) pinned host buffer and copy H2D.
        The copy is enqueued on the current stream; the staging buffer is reused only after an event
        recorded behind its previous copy has completed."""
        total = int(sum(len(w) for w in waves))
        if total and all(isinstance(w, np.ndarray) and w.dtype == np.float32 and w.flags.c_contiguous for w in waves):
            srcs = [torch.from_numpy(w) for w in waves]
            if all(t.is_pinned() for t in srcs):
                # every utterance already sits in page-locked memory (driver: frontend.load_wav_16k(pinned=True)): one async copy each,
                # no packing pass here; torch's host allocator keeps a block from being re-issued until the copy from it has run
                dev = torch.empty(total, dtype=torch.float32, device=self.device)
                o = 0
                for t in srcs:
                    dev[o:o + t.numel()].copy_(t, non_blocking=True)
                    o += t.numel()
                return dev
        pins = self.__dict__.setdefault("_pin_in", {})
        pin, evt = pins.get(slot, (None, None))
        if pin is None or pin.numel() < total:
            pin, evt = torch.empty(max(total, 1 << 20), dtype=torch.float32).pin_memory(), None
        elif evt is not None:
            evt.synchronize()
        host = pin[:total]
        view = host.numpy()
        o = 0
        for w in waves:
            n = len(w)
            view[o:o + n] = w
            o += n
        dev = host.to(self.device, non_blocking=True)
        evt = torch.cuda.Event()
        evt.record()
        pins[slot] = (pin, evt)
        return dev

    def download(self, t: torch.Tensor) -> torch.Tensor:
        """Device fp32 tensor -> fresh pinned host tensor (async D2H + one synchronisation)."""
        host = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return host

    use_tape = True          # replay recorded command lists (one foreign call per forward); False = launch one by one

    @_on_stream
    def forward(self, packed_wave: torch.Tensor, lengths: Sequence[int], slot: int = 0,
                last_state: Optional[int] = None) -> HiddenStates:
        """packed raw samples [sum(lengths)] fp32 on the device -> L+1 hidden states (``last_state`` = N: only states 0..N,
        the launches that feed later states are skipped)."""
        pl = self._plan(lengths, slot)
        last_state = self._check_last_state(last_state)
        flag = self._guard_word(pl)
        if not self.use_tape or self.gemm_trace is not None or self.block_trace is not None:
            self._launches(pl, packed_wave, last_state)      # eager: one Python -> C transition per kernel
        else:
            ar = self._arenas[slot]
            tape = ar.get("tape")
            if tape is None:                                 # first forward in this arena: record, launch nothing
                self._rec = tape = Tape()
                try:
                    self._launches(pl, packed_wave)
                finally:
                    self._rec = None
                ar["tape"] = tape
            tape.inputs["wav"].wav = packed_wave.data_ptr()
            tape.run(pl["sizes"], self._s(), last_state)
        hs = HiddenStates(pl["states"], pl["frame_offs_host"], None if last_state is None else last_state + 1)
        hs.range_flag = flag
        return hs

    def _launches(self, pl, packed_wave: torch.Tensor, last_state: Optional[int] = None) -> None:
        geo = self.geo
        B, M, D, C0 = pl["B"], pl["M"], geo.hidden, geo.conv_dim[0]
        # a6 + a7 (layer 0): zero-mean / unit-variance per utterance fused with framing, then
        # Conv1d(1,C,10,5)+LayerNorm+GELU on the matrix cores (K padded to 64, LayerNorm epilogue)
        fr = pl["frames"]
        rec = self._rec
        a = rec.slot("wave_frames") if rec is not None else _lib.WaveFramesArgs()
        a.wav, a.sample_offs, a.frame_offs = packed_wave.data_ptr(), pl["sample_offs"].data_ptr(), pl["frame_offs0"].data_ptr()
        a.B, a.k, a.stride, a.mode = B, geo.conv_kernel[0], geo.conv_stride[0], self.stem_mode
        a.out, a.out_plane_stride, a.work, a.total_rows = fr.ptr, fr.plane_stride, pl["wave_work"].data_ptr(), pl["rows"][0]
        a.range_flag = self._flag
        if rec is not None:
            rec.inputs["wav"] = a
            rec.commit(_lib.OP_WAVE_FRAMES, a, B=B, total_rows=pl["rows"][0])
        else:
            check(lib.ser_wave_frames_v(C.byref(a), self._s()), "ser_wave_frames")
        a_in = pl["conv_act"][0]
        self._gemm(fr, self.conv0, pl["rows"][0], act=_lib.ACT_GELU, ln=self.conv_ln[0], ln_eps=1e-5, out_act=a_in,
                   k_algo=geo.conv_kernel[0], stem=True)
        # a7: conv layers 1..6 as implicit GEMMs with LayerNorm(C)+GELU fused into the epilogue
        # (the 512-wide output row lives in one block tile, so the pre-LN activations never touch HBM)
        nl = len(geo.conv_dim)
        for i in range(1, nl):
            rows = pl["rows"][i]
            if i < nl - 1:
                a_out = pl["conv_act"][i % 2]
                a_view = Act.__new__(Act)
                a_view.t, a_view.rows, a_view.cols, a_view.planes = a_out.t, rows, C0, a_out.planes
                a_view.plane_stride = a_out.plane_stride
                self._gemm(a_in, self.convs[i - 1], rows, a_rowoff=pl["conv_rowoff"][i - 1], act=_lib.ACT_GELU,
                           ln=self.conv_ln[i], ln_eps=1e-5, out_act=a_view, stem=True)
                a_in = a_view
            else:
                self._gemm(a_in, self.convs[i - 1], rows, a_rowoff=pl["conv_rowoff"][i - 1], act=_lib.ACT_GELU,
                           ln=self.conv_ln[i], ln_eps=1e-5, out_f32=pl["feat_f32"], ldo_f32=C0, stem=True)
        # a9: feature projection (LN -> Linear); also scatter into the zero-halo'd pos-conv input
        self._layernorm(pl["feat_f32"], C0, self.proj_ln, M, C0, out_act=pl["feat_act"], stem=True)
        self._gemm(pl["feat_act"], self.proj, M, out_f32=pl["proj_f32"], ldo_f32=D,
                   out_act=pl["halo_act"], out_rowmap=pl["halo_rowmap"], stem=True,
                   out_mode=self.stem_mode if self.pos_in_stem else self.mode)
        # a10: grouped positional conv + GELU + residual -> hidden_states[0]
        states = pl["states"]
        G, Cg, kc = geo.pos_conv_groups, self.pos_cg, self.pos_kc
        self._gemm(pl["halo_act"], self.pos, M, a_rowoff=pl["pos_rowoff"], kc=kc, ldj=D, groups=G,
                   a_group_stride=Cg, w_group_stride=Cg * geo.pos_conv_kernel * kc, c_group_stride=Cg,
                   N=Cg, K=geo.pos_conv_kernel * kc, act=_lib.ACT_GELU, residual=pl["proj_f32"], ldr=D,
                   out_f32=states[0], ldo_f32=D, k_algo=geo.pos_conv_kernel * Cg, stem=self.pos_in_stem)
        # a11/a12: stable-LayerNorm encoder layers (LayerNorms deferred into the GEMMs)
        self._run_layers(pl, states, pl["first_groups"], B, pl["Tmax"], last_state)


class WhisperEncoder(_EncoderBase):
    """Whisper encoder (log-mel front end + stem convs + pre-LN layers) on libserhip."""

    N_SAMPLES = 480000
    N_FRAMES = 3000

    def __init__(self, geo: EncoderGeometry, state_dict, device="cuda:0", mode: str = "bf16", mel_filters=None):
        super().__init__(geo, device, mode)
        if geo.family != FAMILY_WHISPER:
            raise ValueError("WhisperEncoder needs a whisper geometry")
        sd = state_dict
        D = geo.hidden
        from .frontend import whisper_mel_filters
        self.mel = self._dev_f32(torch.from_numpy(whisper_mel_filters(geo.n_mels) if mel_filters is None else mel_filters))
        w1 = sd["encoder.conv1.weight"].float()
        w2 = sd["encoder.conv2.weight"].float()
        self.conv1 = self._linear(w1.permute(0, 2, 1).reshape(D, -1), sd["encoder.conv1.bias"], stem=True)
        self.conv2 = self._linear(w2.permute(0, 2, 1).reshape(D, -1), sd["encoder.conv2.bias"], stem=True)
        self.pos_emb = self._dev_f32(sd["encoder.embed_positions.weight"])
        self.enc_ln = self._ln_pair(sd, "encoder.layer_norm")
        self.layers = []
        for i in range(geo.num_layers):
            p = f"encoder.layers.{i}"
            self.layers.append(self._layer_weights(
                sd, p, p + ".self_attn", p + ".self_attn_layer_norm", p + ".final_layer_norm",
                p + ".fc1", p + ".fc2", k_bias=False, gate=False, index=i))        # k_proj has no bias

    def _plan(self, lengths, slot: int = 0):
        """Every Whisper shape is a function of B alone (30 s windows), so buffers are keyed by (slot, B) and a new
        batch only uploads its B+1 sample offsets."""
        lengths = tuple(int(n) for n in lengths)
        pl = self._cache.pop((slot, len(lengths)), None)
        if pl is None:
            pl = self._build_plan(len(lengths), slot)
        else:
            self._cache[(slot, len(lengths))] = pl      # re-insert: the dict is kept in least-recently-USED order, the hot
            #                                             full-batch plans of the pipeline slots are never the ones evicted
        if pl["lengths"] != lengths:
            if pl["offs_evt"] is not None:
                pl["offs_evt"].synchronize()
            pl["offs_host"].numpy()[:] = np.concatenate([[0], np.cumsum(lengths)])
            pl["sample_offs"].copy_(pl["offs_host"], non_blocking=True)
            pl["offs_evt"] = torch.cuda.Event()
            pl["offs_evt"].record()
            pl["lengths"] = lengths
        return pl

    def _build_plan(self, B: int, slot: int):
        key_full = (slot, B)
        geo, dev = self.geo, self.device
        D, Fd, nm = geo.hidden, geo.ffn, geo.n_mels
        T2, T1 = geo.max_source_positions, self.N_FRAMES
        if T1 != 2 * T2:
            raise ValueError("max_source_positions must be 1500 (3000 mel frames / 2)")
        Tp = T1 + 2
        M = B * T2
        pl = dict(B=B, M=M, lengths=None, offs_evt=None)
        pl["offs_host"] = torch.empty(B + 1, dtype=torch.int64).pin_memory()
        pl["sample_offs"] = torch.empty(B + 1, dtype=torch.int64, device=dev)
        pl["mel"] = torch.empty((B, nm, T1), dtype=torch.float32, device=dev)
        ws = lib.ser_workspace_bytes(_lib.WS_LOGMEL, B, 0, 0, 0, self.stem_mode)
        pl["work"] = torch.empty(ws, dtype=torch.uint8, device=dev)
        check(lib.ser_logmel_init(pl["work"].data_ptr(), B, _stream()), "ser_logmel_init")     # DFT twiddles: once per buffer
        pl["mel_act"] = self._new_act(B * Tp, nm, stem=True)
        pl["c1_act"] = self._new_act(B * Tp, D, zero=True, stem=True)
        b_idx = np.repeat(np.arange(B, dtype=np.int64), T1)
        t_idx = np.tile(np.arange(T1, dtype=np.int64), B)
        pl["c1_rowoff"] = torch.tensor((b_idx * Tp + t_idx) * nm // 8, dtype=torch.int32, device=dev)
        pl["c1_rowmap"] = torch.tensor(b_idx * Tp + 1 + t_idx, dtype=torch.int32, device=dev)
        b2 = np.repeat(np.arange(B, dtype=np.int64), T2)
        t2 = np.tile(np.arange(T2, dtype=np.int64), B)
        pl["c2_rowoff"] = torch.tensor((b2 * Tp + 2 * t2) * D // 8, dtype=torch.int32, device=dev)
        pl["frame_offs_host"] = [int(b * T2) for b in range(B + 1)]
        pl["frame_offs"] = torch.tensor(pl["frame_offs_host"], dtype=torch.int32, device=dev)
        pl["states"] = torch.empty((geo.num_layers + 1, M, D), dtype=torch.float32, device=dev)
        pl["first_groups"] = 2
        self._layer_buffers(pl, M, pl["first_groups"])
        if len(self._cache) >= 9:                       # three pipeline slots x (full batch, tail batch, retry of one) before anything is evicted
            self._cache.pop(next(iter(self._cache)))
        self._cache[key_full] = pl
        return pl

    upload = SpeechEncoder.upload
    download = SpeechEncoder.download

    def _logmel(self, pl, packed_wave: torch.Tensor) -> None:
        rec = self._rec
        if rec is not None:
            a = rec.slot("logmel")
            a.wav, a.sample_offs, a.mel = packed_wave.data_ptr(), pl["sample_offs"].data_ptr(), self.mel.data_ptr()
            a.out, a.work, a.B, a.n_mels = pl["mel"].data_ptr(), pl["work"].data_ptr(), pl["B"], self.geo.n_mels
            rec.inputs["wav"] = a
            rec.commit(_lib.OP_LOGMEL, a)
            return
        check(lib.ser_logmel_whisper(packed_wave.data_ptr(), pl["sample_offs"].data_ptr(), pl["B"], self.mel.data_ptr(),
                                     self.geo.n_mels, pl["mel"].data_ptr(), pl["work"].data_ptr(), self._s()),
              "ser_logmel_whisper")

    @_on_stream
    def log_mel(self, packed_wave: torch.Tensor, lengths: Sequence[int], slot: int = 0) -> torch.Tensor:
        """a16: [B, n_mels, 3000] fp32 input_features, computed on the GPU."""
        pl = self._plan(lengths, slot)
        self._logmel(pl, packed_wave)
        return pl["mel"]

    use_tape = True          # replay a recorded command list (one foreign call per forward), like SpeechEncoder

    @_on_stream
    def forward(self, packed_wave: torch.Tensor, lengths: Sequence[int], slot: int = 0,
                last_state: Optional[int] = None) -> HiddenStates:
        """packed raw samples -> log-mel (a16) -> encoder (a17): the reference's processor + model.encoder calls
        (preprocess_whisper.py:48,57).  Every buffer is a function of (slot, B), so the ~170 launches are recorded once
        per plan and replayed with one ser_run call; only the waveform pointer changes from batch to batch.
        ``last_state`` = N stops after hidden state N (``--n_layer N``, preprocess_whisper.py:71)."""
        pl = self._plan(lengths, slot)
        last_state = self._check_last_state(last_state)
        flag = self._guard_word(pl)
        if not self.use_tape or self.gemm_trace is not None or self.block_trace is not None:
            self._logmel(pl, packed_wave)
            self._encoder_launches(pl, pl["mel"], last_state)
        else:
            tape = pl.get("tape")
            if tape is None:
                self._rec = tape = Tape()
                try:
                    self._logmel(pl, packed_wave)
                    self._encoder_launches(pl, pl["mel"])
                finally:
                    self._rec = None
                pl["tape"] = tape
            tape.inputs["wav"].wav = packed_wave.data_ptr()
            tape.run({}, self._s(), last_state)
        hs = HiddenStates(pl["states"], pl["frame_offs_host"], None if last_state is None else last_state + 1)
        hs.range_flag = flag
        return hs

    @_on_stream
    def forward_features(self, input_features: torch.Tensor, lengths: Sequence[int], slot: int = 0) -> HiddenStates:
        """a17: ``model.encoder(input_features, output_hidden_states=True).hidden_states`` on caller-supplied features."""
        geo = self.geo
        pl = self._plan(lengths, slot)
        if tuple(input_features.shape) != (pl["B"], geo.n_mels, self.N_FRAMES):
            raise ValueError(f"Whisper expects input_features of shape {(pl['B'], geo.n_mels, self.N_FRAMES)}, "
                             f"got {tuple(input_features.shape)}")
        flag = self._guard_word(pl)
        self._encoder_launches(pl, input_features)
        hs = HiddenStates(pl["states"], pl["frame_offs_host"])
        hs.range_flag = flag
        return hs

    def _encoder_launches(self, pl, input_features: torch.Tensor, last_state: Optional[int] = None) -> None:
        geo = self.geo
        B, M, D, nm = pl["B"], pl["M"], geo.hidden, geo.n_mels
        T1, T2 = self.N_FRAMES, geo.max_source_positions
        ma = pl["mel_act"]
        rec = self._rec
        a = rec.slot("pack_act") if rec is not None else _lib.PackActArgs()
        a.x, a.out, a.ldo, a.out_plane_stride = input_features.data_ptr(), ma.ptr, nm, ma.plane_stride
        a.B, a.C, a.T, a.halo, a.mode = B, nm, T1, 1, self.stem_mode
        a.range_flag = self._flag
        if rec is not None:
            rec.commit(_lib.OP_PACK_ACT, a)
        else:
            check(lib.ser_pack_act_v(C.byref(a), self._s()), "ser_pack_act")
        # stem: gelu(conv1 k3 p1), gelu(conv2 k3 s2 p1) + embed_positions -> hidden_states[0]
        self._gemm(ma, self.conv1, B * T1, a_rowoff=pl["c1_rowoff"], act=_lib.ACT_GELU, out_act=pl["c1_act"],
                   out_rowmap=pl["c1_rowmap"], stem=True)
        states = pl["states"]
        self._gemm(pl["c1_act"], self.conv2, M, a_rowoff=pl["c2_rowoff"], act=_lib.ACT_GELU, residual=self.pos_emb,
                   ldr=D, res_row_mod=T2, out_f32=states[0], ldo_f32=D, stem=True)
        self._run_layers(pl, states, pl["first_groups"], B, T2, last_state)


class TextEncoder(_EncoderBase):
    """RoBERTa text encoder (next row 8f-1: preprocessing/preprocess_roberta.py) on the same kernels:
    embeddings + L post-LayerNorm BERT layers.  ``forward(input_ids [B,T], attention_mask [B,T])`` returns
    L+1 states per sequence, ALL T rows each (the reference saves the padded positions too); padded KEYS
    are excluded through per-sequence key lengths (tokenizer padding="max_length" pads on the right)."""

    def __init__(self, geo: EncoderGeometry, state_dict, device="cuda:0", mode: str = "bf16"):
        super().__init__(geo, device, mode)
        if geo.family != FAMILY_ROBERTA:
            raise ValueError("TextEncoder needs a roberta geometry")
        if mode in ("f16", "f16q", "f16a", "f16m", "f16mf"):
            raise ValueError("the text encoders support the bf16, fp32x and f16x numerics modes")
        sd = state_dict
        D = geo.hidden
        self.wemb = self._dev_f32(sd["embeddings.word_embeddings.weight"])
        self.pemb = self._dev_f32(sd["embeddings.position_embeddings.weight"])
        self.temb = self._dev_f32(sd["embeddings.token_type_embeddings.weight"][0])
        self.emb_ln = self._ln_pair(sd, "embeddings.LayerNorm")
        self.layers = []
        for i in range(geo.num_layers):
            p = f"encoder.layer.{i}"
            a = p + ".attention.self"
            qkv_w = torch.cat([sd[a + ".query.weight"], sd[a + ".key.weight"], sd[a + ".value.weight"]], 0)
            qkv_b = torch.cat([sd[a + ".query.bias"], sd[a + ".key.bias"], sd[a + ".value.bias"]], 0)
            self.layers.append(dict(
                qkv=self._linear(qkv_w, qkv_b),
                out=self._linear(sd[p + ".attention.output.dense.weight"], sd[p + ".attention.output.dense.bias"]),
                ln1=self._ln_pair(sd, p + ".attention.output.LayerNorm"),
                fc1=self._linear(sd[p + ".intermediate.dense.weight"], sd[p + ".intermediate.dense.bias"]),
                fc2=self._linear(sd[p + ".output.dense.weight"], sd[p + ".output.dense.bias"]),
                ln2=self._ln_pair(sd, p + ".output.LayerNorm")))

    def _plan(self, B: int, T: int, slot: int = 0):
        key = (slot, B, T)
        if key in self._cache:
            return self._cache[key]
        geo, dev = self.geo, self.device
        D, Fd, M = geo.hidden, geo.ffn, B * T
        pl = dict(B=B, T=T, M=M)
        pl["frame_offs_host"] = [b * T for b in range(B + 1)]
        pl["frame_offs"] = torch.tensor(pl["frame_offs_host"], dtype=torch.int32, device=dev)
        pl["states"] = torch.empty((geo.num_layers + 1, M, D), dtype=torch.float32, device=dev)
        pl["xa"], pl["ha"] = self._new_act(M, D), self._new_act(M, D)
        pl["qkv"], pl["ctx"], pl["ffn"] = self._new_act(M, 3 * D), self._new_act(M, D), self._new_act(M, Fd)
        pl["tmp"] = torch.empty((M, D), dtype=torch.float32, device=dev)
        pl["h"] = torch.empty((M, D), dtype=torch.float32, device=dev)
        if len(self._cache) >= 4:
            self._cache.pop(next(iter(self._cache)))
        self._cache[key] = pl
        return pl

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, slot: int = 0) -> HiddenStates:
        """``model(input_ids, attention_mask, output_hidden_states=True).hidden_states`` (preprocess_roberta.py:57,68)."""
        geo = self.geo
        B, T = input_ids.shape
        if attention_mask.shape != input_ids.shape:
            raise ValueError("attention_mask must have the shape of input_ids")
        if T + geo.pad_token_id + 1 > geo.max_positions:
            # position ids run up to T + padding_idx (HF modeling_roberta.py:142): the table must hold them
            raise ValueError(f"{T} tokens need {T + geo.pad_token_id + 1} position embeddings, the model has {geo.max_positions}")
        if int(input_ids.min()) < 0 or int(input_ids.max()) >= geo.vocab_size:
            raise ValueError("token id outside the vocabulary")
        mask = attention_mask.to(torch.int64).cpu()
        klen = mask.sum(dim=1)
        if not torch.equal(mask, (torch.arange(T)[None, :] < klen[:, None]).to(torch.int64)) or int(klen.min()) < 1:
            raise ValueError("attention_mask must be right-padded with at least one valid token per sequence")
        ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
        key_lens = klen.to(device=self.device, dtype=torch.int32)
        return self.forward_device(ids, key_lens, slot)

    @_on_stream
    def forward_device(self, ids: torch.Tensor, key_lens: torch.Tensor, slot: int = 0) -> HiddenStates:
        """Same as ``forward`` with inputs already on the device (int32 ids [B,T], int32 key lengths [B]):
        no host synchronisation, so it can be captured into a hipGraph."""
        geo = self.geo
        B, T = ids.shape
        pl = self._plan(B, T, slot)
        flag = self._guard_word(pl)
        M, D = pl["M"], geo.hidden
        states = pl["states"]
        xa = pl["xa"]
        check(lib.ser_embed_ln(ids.data_ptr(), self.wemb.data_ptr(), self.pemb.data_ptr(), self.temb.data_ptr(),
                               self.emb_ln[0].data_ptr(), self.emb_ln[1].data_ptr(), float(geo.layer_norm_eps),
                               states[0].data_ptr(), xa.ptr, xa.plane_stride, self.mode, B, T, D, geo.pad_token_id,
                               _stream()), "ser_embed_ln")
        scale = geo.head_dim ** -0.5 * 1.4426950408889634
        for i, lay in enumerate(self.layers):
            x = states[i]
            self._gemm(xa, lay["qkv"], M, out_act=pl["qkv"], col_scale=scale, col_scale_end=D)
            self._attention(pl["qkv"], pl["frame_offs"], B, T, pl["ctx"], key_lens=key_lens)
            self._gemm(pl["ctx"], lay["out"], M, residual=x, ldr=D, out_f32=pl["tmp"], ldo_f32=D)
            self._layernorm(pl["tmp"], D, lay["ln1"], M, D, out_f32=pl["h"], out_act=pl["ha"])
            self._gemm(pl["ha"], lay["fc1"], M, act=_lib.ACT_GELU, out_act=pl["ffn"])
            self._gemm(pl["ffn"], lay["fc2"], M, residual=pl["h"], ldr=D, out_f32=pl["tmp"], ldo_f32=D)
            self._layernorm(pl["tmp"], D, lay["ln2"], M, D, out_f32=states[i + 1], out_act=xa)
        hs = HiddenStates(states, pl["frame_offs_host"])
        hs.range_flag = flag
        return hs


def _deberta_log_bucket(rel: torch.Tensor, bucket_size: int, max_position: int) -> torch.Tensor:
    """Signed distance -> bucket, in the float32 arithmetic HF uses (modeling_deberta_v2.py make_log_bucket_position):
    identity inside +-bucket_size/2, logarithmic beyond.  O(T) host work per sequence length; the kernels only see the
    resulting integer columns."""
    mid = bucket_size // 2
    sign = torch.sign(rel)
    inside = (rel < mid) & (rel > -mid)
    a = torch.where(inside, torch.full_like(rel, mid - 1), rel.abs()).to(torch.float32)
    logp = torch.ceil(torch.log(a / mid) / math.log((max_position - 1) / mid) * (mid - 1)) + mid
    return torch.where(a <= mid, rel.to(torch.float32), logp * sign).to(torch.long)


class DebertaEncoder(_EncoderBase):
    """DeBERTa-v2/v3 text encoder (preprocessing/preprocess_deroberta.py builds it with AutoModel) in its v3
    configuration: LayerNorm-ed word embeddings (no absolute positions / token types), L post-LayerNorm blocks with
    disentangled attention.  The relative-position side is input independent, so it is folded at load: LayerNorm of the
    relative embeddings and their projection through every layer's (shared) query / key weights.  Per layer the device
    then runs the packed QKV GEMM, two grouped GEMMs (content queries x position keys, content keys x position queries,
    restricted to the relative rows a T-token sequence can reach), ``ser_deberta_attention``, and the same post-LN
    output / feed-forward GEMMs as the RoBERTa encoder.  Same call surface as ``TextEncoder``."""

    def __init__(self, geo: EncoderGeometry, state_dict, device="cuda:0", mode: str = "bf16"):
        super().__init__(geo, device, mode)
        if geo.family != "deberta":
            raise ValueError("DebertaEncoder needs a deberta geometry")
        if mode in ("f16", "f16q", "f16a", "f16m", "f16mf"):
            raise ValueError("the text encoders support the bf16, fp32x and f16x numerics modes")
        if geo.head_dim != 64:
            raise ValueError("DeBERTa path: head dim must be 64 (K of the position GEMMs; deberta-v3 base/large have 64)")
        sd = state_dict
        D, span = geo.hidden, geo.position_buckets
        self.wemb = self._dev_f32(sd["embeddings.word_embeddings.weight"])
        self.emb_ln = self._ln_pair(sd, "embeddings.LayerNorm")
        rel = sd["encoder.rel_embeddings.weight"][: 2 * span].double()
        rel = torch.nn.functional.layer_norm(rel, (D,), sd["encoder.LayerNorm.weight"].double(), sd["encoder.LayerNorm.bias"].double(),
                                             geo.layer_norm_eps)
        self.layers = []
        for i in range(geo.num_layers):
            p = f"encoder.layer.{i}"
            a = p + ".attention.self"
            wq, wk = sd[a + ".query_proj.weight"].double(), sd[a + ".key_proj.weight"].double()
            bq, bk = sd[a + ".query_proj.bias"].double(), sd[a + ".key_proj.bias"].double()
            qkv_w = torch.cat([sd[a + ".query_proj.weight"], sd[a + ".key_proj.weight"], sd[a + ".value_proj.weight"]], 0)
            qkv_b = torch.cat([sd[a + ".query_proj.bias"], sd[a + ".key_proj.bias"], sd[a + ".value_proj.bias"]], 0)
            self.layers.append(dict(
                qkv=self._linear(qkv_w, qkv_b),
                pos_q=(rel @ wq.T + bq).float(), pos_k=(rel @ wk.T + bk).float(),       # [2 span, D] fp32, host: windows cut per T
                out=self._linear(sd[p + ".attention.output.dense.weight"], sd[p + ".attention.output.dense.bias"]),
                ln1=self._ln_pair(sd, p + ".attention.output.LayerNorm"),
                fc1=self._linear(sd[p + ".intermediate.dense.weight"], sd[p + ".intermediate.dense.bias"]),
                fc2=self._linear(sd[p + ".output.dense.weight"], sd[p + ".output.dense.bias"]),
                ln2=self._ln_pair(sd, p + ".output.LayerNorm")))
        # deberta-v2 xlarge / xxlarge: ConvLayer after layer 0 (HF modeling_deberta_v2.py ConvLayer) as an implicit-conv GEMM
        self.text_conv = None
        if geo.text_conv_kernel:
            if geo.text_conv_kernel != 3:
                raise ValueError("DeBERTa ConvLayer: kernel size 3 (deberta-v2-xlarge / xxlarge) is what is built")
            wc = sd["encoder.conv.conv.weight"].float()                                    # [out, in, tap]
            self.text_conv = dict(lin=self._linear(wc.permute(0, 2, 1).reshape(D, 3 * D), sd["encoder.conv.conv.bias"]),
                                  ln=self._ln_pair(sd, "encoder.conv.LayerNorm"))
        self._windows: Dict[int, dict] = {}

    def _window(self, T: int):
        """Everything that depends on the sequence length only: which relative rows are reachable, the column of every
        signed distance inside that window, and each layer's position keys / queries cut to it as [H, Nr, dh] GEMM weights."""
        if T in self._windows:
            return self._windows[T]
        geo = self.geo
        span, H, dh = geo.position_buckets, geo.heads, geo.head_dim
        d = torch.arange(-(T - 1), T)
        bucket = _deberta_log_bucket(d, span, geo.max_positions)
        c2p_row = torch.clamp(bucket + span, 0, 2 * span - 1)            # row of pos_k for distance d = q - k
        p2c_row = torch.clamp(-bucket + span, 0, 2 * span - 1)           # row of pos_q for distance d = k - q
        lo = int(min(c2p_row.min(), p2c_row.min()))
        hi = int(max(c2p_row.max(), p2c_row.max())) + 1
        Nr = min(2 * span, (hi - lo + 7) // 8 * 8)                       # GEMM N: a multiple of 8 (2 span is one)
        lo = max(0, min(lo, 2 * span - Nr))                              # ... so slide the window back inside the table
        w = dict(Nr=Nr, lo=lo,
                 c2p_col=(c2p_row - lo).to(torch.int32).to(self.device), p2c_col=(p2c_row - lo).to(torch.int32).to(self.device), layers=[])
        for lay in self.layers:
            cut = lambda m: m[lo:lo + Nr].reshape(Nr, H, dh).permute(1, 0, 2).reshape(H * Nr, dh)     # noqa: E731
            # ONE grouped GEMM gives both position terms: the k columns follow the q columns in the packed projection, so
            # group g < H reads query head g and group H + h reads key head h with the same column step dh; the weights are
            # the position keys of every head followed by the position queries of every head
            w["layers"].append(self._linear(torch.cat([cut(lay["pos_k"]), cut(lay["pos_q"])], 0), None))
        self._windows[T] = w
        return w

    def _plan(self, B: int, T: int, slot: int = 0):
        key = (slot, B, T)
        if key in self._cache:
            return self._cache[key]
        geo, dev = self.geo, self.device
        D, Fd, M = geo.hidden, geo.ffn, B * T
        win = self._window(T)
        pl = dict(B=B, T=T, M=M, win=win)
        pl["frame_offs_host"] = [b * T for b in range(B + 1)]
        pl["states"] = torch.empty((geo.num_layers + 1, M, D), dtype=torch.float32, device=dev)
        pl["xa"], pl["ha"] = self._new_act(M, D), self._new_act(M, D)
        pl["qkv"], pl["ctx"], pl["ffn"] = self._new_act(M, 3 * D), self._new_act(M, D), self._new_act(M, Fd)
        pl["posterms"] = torch.empty((M, 2 * geo.heads * win["Nr"]), dtype=torch.float32, device=dev)   # [c2p of every head | p2c of every head]
        pl["frame_offs"] = torch.tensor(pl["frame_offs_host"], dtype=torch.int32, device=dev)
        pl["bias2d"] = torch.zeros((B, geo.heads, T, (T + 63) // 64 * 64), dtype=torch.float32, device=dev)   # dense c2p + p2c bias
        if self.text_conv is not None:
            pl["conv_in"] = self._new_act(B * (T + 2), D, zero=True)                       # [0][tokens of sequence b][0] per sequence
            b_idx = np.repeat(np.arange(B, dtype=np.int64), T)
            t_idx = np.tile(np.arange(T, dtype=np.int64), B)
            pl["conv_rowoff"] = torch.tensor((b_idx * (T + 2) + t_idx) * D // 8, dtype=torch.int32, device=dev)   # tap 0 = token t-1
        pl["tmp"] = torch.empty((M, D), dtype=torch.float32, device=dev)
        pl["h"] = torch.empty((M, D), dtype=torch.float32, device=dev)
        if len(self._cache) >= 4:
            self._cache.pop(next(iter(self._cache)))
        self._cache[key] = pl
        return pl

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, slot: int = 0) -> HiddenStates:
        """``model(input_ids, attention_mask, output_hidden_states=True).hidden_states`` (preprocess_deroberta.py:57,68)."""
        B, T = input_ids.shape
        if attention_mask.shape != input_ids.shape:
            raise ValueError("attention_mask must have the shape of input_ids")
        if int(input_ids.min()) < 0 or int(input_ids.max()) >= self.geo.vocab_size:
            raise ValueError("token id outside the vocabulary")
        mask = attention_mask.to(torch.int64).cpu()
        klen = mask.sum(dim=1)
        if not torch.equal(mask, (torch.arange(T)[None, :] < klen[:, None]).to(torch.int64)) or int(klen.min()) < 1:
            raise ValueError("attention_mask must be right-padded with at least one valid token per sequence")
        ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
        return self.forward_device(ids, klen.to(device=self.device, dtype=torch.int32), slot)

    @_on_stream
    def forward_device(self, ids: torch.Tensor, key_lens: torch.Tensor, slot: int = 0) -> HiddenStates:
        geo = self.geo
        B, T = ids.shape
        pl = self._plan(B, T, slot)
        flag = self._guard_word(pl)
        M, D, H, dh = pl["M"], geo.hidden, geo.heads, geo.head_dim
        win = pl["win"]
        Nr = win["Nr"]
        states, xa, qkv = pl["states"], pl["xa"], pl["qkv"]
        st = self._s()
        check(lib.ser_embed_ln_masked(ids.data_ptr(), self.wemb.data_ptr(), self.emb_ln[0].data_ptr(), self.emb_ln[1].data_ptr(),
                                      float(geo.layer_norm_eps), key_lens.data_ptr(), states[0].data_ptr(), xa.ptr, xa.plane_stride,
                                      self.mode, B, T, D, st), "ser_embed_ln_masked")
        # scores run in the exp2 domain: q leaves the projection multiplied by (3 dh)^-0.5 * log2(e), so the content term and the
        # content -> position term (a product with q) arrive scaled; the position -> content term (a product with k) is scaled
        # by ser_deberta_bias.  (HF: (Qc Kc^T + c2p + p2c) / sqrt(3 dh), modeling_deberta_v2.py DisentangledSelfAttention.)
        s2 = (3.0 * dh) ** -0.5 * 1.4426950408889634
        bias2d = pl["bias2d"]
        for i, lay in enumerate(self.layers):
            x = states[i]
            self._gemm(xa, lay["qkv"], M, out_act=qkv, col_scale=s2, col_scale_end=D)
            # content -> position (q x position keys) and position -> content (k x position queries) terms: one grouped GEMM,
            # group = (term, head), K = dh
            pt = pl["posterms"]
            self._gemm(qkv, win["layers"][i], M, groups=2 * H, a_group_stride=dh, w_group_stride=Nr * dh, c_group_stride=Nr,
                       N=Nr, K=dh, out_f32=pt, ldo_f32=2 * H * Nr)
            # dense bias of every (sequence, head) from the two gathers, then the matrix-core attention kernel of the speech
            # encoders with it (the 80-token VALU kernel ser_deberta_attention was a third of the step)
            check(lib.ser_deberta_bias(pt.data_ptr(), pt.data_ptr() + 4 * H * Nr, 2 * H * Nr, Nr, win["c2p_col"].data_ptr(),
                                       win["p2c_col"].data_ptr(), key_lens.data_ptr(), bias2d.data_ptr(), bias2d.shape[-1],
                                       B, T, H, float(s2), st), "ser_deberta_bias")
            self._attention(qkv, pl["frame_offs"], B, T, pl["ctx"], key_lens=key_lens, bias2d=bias2d)
            self._gemm(pl["ctx"], lay["out"], M, residual=x, ldr=D, out_f32=pl["tmp"], ldo_f32=D)
            self._layernorm(pl["tmp"], D, lay["ln1"], M, D, out_f32=pl["h"], out_act=pl["ha"])
            self._gemm(pl["ha"], lay["fc1"], M, act=_lib.ACT_GELU, out_act=pl["ffn"])
            self._gemm(pl["ffn"], lay["fc2"], M, residual=pl["h"], ldr=D, out_f32=pl["tmp"], ldo_f32=D)
            self._layernorm(pl["tmp"], D, lay["ln2"], M, D, out_f32=states[i + 1], out_act=xa)
            if i == 0 and self.text_conv is not None:
                # ConvLayer: LN(layer-0 output + gelu(Conv1d_k3(embeddings))) with padded rows zero.  (HF zeroes the padded
                # rows of the conv output before the activation too; gelu(0) = 0 and those rows are zeroed at the end anyway.)
                ci = pl["conv_in"]
                check(lib.ser_pack_rows(states[0].data_ptr(), D, B, T, D, 1, ci.ptr, D, ci.plane_stride, self.mode, st), "ser_pack_rows")
                self._gemm(ci, self.text_conv["lin"], M, a_rowoff=pl["conv_rowoff"], kc=D, ldj=D, K=3 * D, act=_lib.ACT_GELU,
                           residual=states[1], ldr=D, out_f32=pl["tmp"], ldo_f32=D)
                self._layernorm(pl["tmp"], D, self.text_conv["ln"], M, D, out_f32=states[1], out_act=xa)
                check(lib.ser_zero_padded_rows(states[1].data_ptr(), D, xa.ptr, D, xa.plane_stride, self.mode, key_lens.data_ptr(),
                                               B, T, D, st), "ser_zero_padded_rows")
        hs = HiddenStates(states, pl["frame_offs_host"])
        hs.range_flag = flag
        return hs


def mean_last4(hs: HiddenStates) -> torch.Tensor:
    """``--use_average y``: mean of the last four states (preprocess_speech.py:52-63), on the GPU."""
    s = hs.states
    if getattr(hs, "computed", s.shape[0]) < s.shape[0]:
        raise IndexError("mean of the last four states needs the full forward (this one stopped early: last_state)")
    out = torch.empty_like(s[0])
    n = out.numel()
    check(lib.ser_mean4(s[-4].data_ptr(), s[-3].data_ptr(), s[-2].data_ptr(), s[-1].data_ptr(), out.data_ptr(), n, _stream()),
          "ser_mean4")
    return out


def build_encoder(geo: EncoderGeometry, state_dict, device="cuda:0", mode="bf16"):
    if geo.family == "deberta":
        return DebertaEncoder(geo, state_dict, device, mode)
    if geo.family == FAMILY_ROBERTA:
        return TextEncoder(geo, state_dict, device, mode)
    if geo.family == FAMILY_WHISPER:
        return WhisperEncoder(geo, state_dict, device, mode)
    return SpeechEncoder(geo, state_dict, device, mode)

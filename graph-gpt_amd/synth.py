"""Seeded synthetic Eulerian-token batches with the reference's input contract.

Build-side counterpart of the host collator + SMTP masking:
  * `DataCollatorForGST.__call__` (reference src/data/collator.py:70-111): right padding with 0 to
    S = 8*ceil(max_len/8), attention_mask 1/0, position_ids 0..len-1 (0 on pads);
  * `prepare_inputs_for_pretrain_mlm` + `_mask_stacked_input_ids_v2`
    (reference src/utils/tokenizer_utils.py:222-363, :112-148): per sample t = umr_min +
    (umr_max-umr_min)*U, alpha = 1 - t**power, ceil(alpha*len*F) cells of the sample's own len*F
    cells chosen uniformly, masked cell -> <mask> id 1, labels = original id at masked cells else -100,
    optional dLM weight power/t.
Token ids are drawn from [first_id, V) which is arithmetically equivalent to the real vocab ranges
(SURVEY.md §8d).  NumPy RNG => identical batches here and on the GPU box.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

MASK_TOKEN_ID = 1
PAD_TOKEN_ID = 0
LABEL_PAD = -100


def _lengths(rng, B, S, mode: str, min_len: int, mean_len: float = 22.0):
    if mode == "full":
        return np.full(B, S, np.int64)
    if mode == "uniform":
        return rng.randint(min_len, S + 1, size=B).astype(np.int64)
    if mode == "pcqm":  # clipped N(22, 6), min 6
        ln = np.rint(rng.normal(mean_len, 6.0, size=B)).astype(np.int64)
        return np.clip(ln, max(6, min_len), S)
    raise ValueError(mode)


def make_pretrain_batch(B: int, S: int, F: int, V: int, seed: int = 1234, *, lengths: str = "pcqm",
                        min_len: int = 4, first_id: int = 22, power: float = 1.0,
                        umr_clip=(0.01, 0.99), dlm_wgt: bool = False, force_full_row: bool = True,
                        long_tail: float = 0.0, mean_len: float = 22.0) -> Dict[str, np.ndarray]:
    """SMTP pre-train batch: input_ids/labels i64 [B,S,F], attention_mask/position_ids i64 [B,S]."""
    assert S % 8 == 0 or True  # collator pads to a multiple of 8; callers choose S accordingly
    rng = np.random.RandomState(seed)
    lens = _lengths(rng, B, S, lengths, min_len, mean_len)
    if force_full_row and lengths != "full":
        lens[rng.randint(B)] = S  # the batch max defines S in the real collator
    if long_tail > 0.0 and S > 32:   # a heavier long tail (bench.py --long-tail): own generator, the other draws stay those of long_tail = 0
        rt = np.random.RandomState(seed ^ 0x5A17)
        pick = rt.random_sample(B) < long_tail
        lens[pick] = rt.randint(33, S + 1, size=int(pick.sum()))
    ids = np.zeros((B, S, F), np.int64)
    labels = np.full((B, S, F), LABEL_PAD, np.int64)
    att = np.zeros((B, S), np.int64)
    pos = np.zeros((B, S), np.int64)
    wgt = np.zeros((B,), np.float32)
    lo = min(first_id, V - 1)
    for b in range(B):
        n = int(lens[b])
        tok = rng.randint(lo, V, size=(n, F)).astype(np.int64)
        t = umr_clip[0] + (umr_clip[1] - umr_clip[0]) * rng.random_sample()
        alpha = 1.0 - t ** power
        wgt[b] = power / t
        k = int(np.ceil(n * F * alpha))
        flat = rng.permutation(n * F)[:k]
        lab = np.full((n * F,), LABEL_PAD, np.int64)
        flat_tok = tok.reshape(-1)
        lab[flat] = flat_tok[flat]
        flat_tok[flat] = MASK_TOKEN_ID
        ids[b, :n] = flat_tok.reshape(n, F)
        labels[b, :n] = lab.reshape(n, F)
        att[b, :n] = 1
        pos[b, :n] = np.arange(n)
    out = dict(input_ids=ids, labels=labels, attention_mask=att, position_ids=pos, lengths=lens)
    if dlm_wgt:
        out["wgt"] = wgt
    return out


def make_packed_pretrain_batch(B: int, S: int, F: int, V: int, seed: int = 1234, *, mean_len: int = 12, min_len: int = 4,
                               first_id: int = 22, eos_id: int = 3) -> Dict[str, np.ndarray]:
    """Token packing (reference GraphsMapDataset.pack_token_seq, src/data/tokenizer.py:359-415, and
    prepare_inputs_for_pretrain_mlm, src/utils/tokenizer_utils.py:340-355): every row holds several graphs back to back,
    separated by one all-<eos> position, until the row is (nearly) full; position_ids run on across graphs, the attention
    mask is the 3-D block-diagonal matrix [B,S,S] (bi-directional inside a graph, the separator belongs to the graph it
    closes), rows beyond the packed length are zero.  Masking as in make_pretrain_batch, per packed row."""
    rng = np.random.RandomState(seed)
    ids = np.zeros((B, S, F), np.int64)
    labels = np.full((B, S, F), LABEL_PAD, np.int64)
    att = np.zeros((B, S, S), np.int64)
    pos = np.zeros((B, S), np.int64)
    seg = np.full((B, S), -1, np.int64)
    lo = min(first_id, V - 1)
    used = np.zeros((B,), np.int64)
    for b in range(B):
        n, g = 0, 0
        while True:
            ln = int(np.clip(rng.normal(mean_len, mean_len / 3), min_len, S))
            if n + ln + 1 > S - (0 if b % 2 else 3):     # every other row keeps a few pad positions at the end
                break
            tok = rng.randint(lo, V, size=(ln, F)).astype(np.int64)
            ids[b, n:n + ln] = tok
            ids[b, n + ln] = eos_id
            att[b, n:n + ln + 1, n:n + ln + 1] = 1
            seg[b, n:n + ln + 1] = g
            n += ln + 1
            g += 1
        used[b] = n
        pos[b, :n] = np.arange(n)
        t = 0.01 + 0.98 * rng.random_sample()
        k = int(np.ceil(n * F * (1.0 - t)))
        flat = rng.permutation(n * F)[:k]
        flat_tok = ids[b, :n].reshape(-1).copy()
        lab = np.full((n * F,), LABEL_PAD, np.int64)
        lab[flat] = flat_tok[flat]
        flat_tok[flat] = MASK_TOKEN_ID
        ids[b, :n] = flat_tok.reshape(n, F)
        labels[b, :n] = lab.reshape(n, F)
    return dict(input_ids=ids, labels=labels, attention_mask=att, position_ids=pos, lengths=used, segments=seg)


def make_task_batch(B: int, S: int, F: int, V: int, seed: int = 1234, *, lengths: str = "uniform",
                    min_len: int = 8, first_id: int = 22, num_labels: int = 2,
                    regression: bool = False, multi_label: bool = False) -> Dict[str, np.ndarray]:
    """Fine-tune batch (edge/graph-level): ids, attention_mask, position_ids, task_labels [B]
    (multi_label: float [B, num_labels] in {0, 1} with NaN = unlabelled, the ogbg-molpcba convention the reference's
    `is_labeled = labels == labels` relies on, modeling_finetune.py:227-230)."""
    rng = np.random.RandomState(seed)
    lens = _lengths(rng, B, S, lengths, min_len)
    if lengths != "full":
        lens[rng.randint(B)] = S
    ids = np.zeros((B, S, F), np.int64)
    att = np.zeros((B, S), np.int64)
    pos = np.zeros((B, S), np.int64)
    lo = min(first_id, V - 1)
    for b in range(B):
        n = int(lens[b])
        ids[b, :n] = rng.randint(lo, V, size=(n, F))
        att[b, :n] = 1
        pos[b, :n] = np.arange(n)
    if regression:
        y = rng.standard_normal(size=(B,)).astype(np.float32)
    elif multi_label:
        y = rng.randint(0, 2, size=(B, num_labels)).astype(np.float32)
        y[rng.rand(B, num_labels) < 0.25] = np.nan
        y[0, 0] = 1.0   # at least one labelled entry
    else:
        y = rng.randint(0, num_labels, size=(B,)).astype(np.int64)
    return dict(input_ids=ids, attention_mask=att, position_ids=pos, task_labels=y, lengths=lens)


def real_tokens(batch: Dict[str, np.ndarray]) -> int:
    """Un-padded graph tokens in the batch = the unit of the headline metric
    (reference src/conf/stats_configs.py:69-76, src/utils/misc_utils.py:349-378)."""
    return int(batch["attention_mask"].sum())

"""Deterministic stand-ins for the CLIP tokenizer / text encoder (no vocabulary or checkpoints in the
sandbox): whitespace tokenizer with the Hugging Face call surface the reference uses
(models/models.py:63-89, models/pipelines.py:303-304, utils/guidance.py:10-89) and an embedding-table "encoder".
Used on both sides of the orchestration goldens: by oracle/make_golden_runs.py (driving the reference's own
generation/*.run on CPU) and by the GPU tests (driving the drop-in plugins)."""
import zlib

import numpy as np
import torch


class _Tok(dict):
    def __getattr__(self, k):
        return self[k]

    def to(self, *_a, **_k):
        return self


class FakeTokenizer:
    model_max_length = 77
    eos_token = "<eos>"

    def __init__(self):
        self.vocab, self.rev = {"<bos>": 0, "<eos>": 1}, {0: "<bos>", 1: "<eos>"}

    def _id(self, w):
        """A word's id is a stable hash of the word (not its order of first appearance): the embeddings of a
        prompt must not depend on which prompts were tokenised before it."""
        if w not in self.vocab:
            i = 2 + zlib.crc32(w.encode()) % 2000003
            while i in self.rev and self.rev[i] != w:
                i += 1
            self.vocab[w] = i
            self.rev[i] = w
        return self.vocab[w]

    def _convert_id_to_token(self, i):
        return self.rev[int(i)]

    def __call__(self, texts, padding="do_not_pad", max_length=77, truncation=False, return_tensors="pt"):
        rows = [[0] + [self._id(w) for w in t.replace(",", " ,").split()][:75] + [1] for t in texts]
        if padding == "max_length":
            rows = [r + [1] * (max_length - len(r)) for r in rows]
        elif padding is True:
            m = max(len(r) for r in rows)
            rows = [r + [1] * (m - len(r)) for r in rows]
        if return_tensors == "np":
            return _Tok(input_ids=[np.array(r) for r in rows])
        return _Tok(input_ids=torch.tensor(rows))


class FakeTextEncoder:
    """ids -> rows of a seeded random table (hidden states) and their mean over tokens at width 768 (pooler_output).
    The table depends on the token id only, so the same prompt gives the same embeddings on any device."""

    def __init__(self, cx, device="cuda"):
        self.cx, self.device = cx, device

    def _emb(self, ids, dim):
        g = torch.Generator().manual_seed(1234)
        table = torch.randn(4096, dim, generator=g)
        return table[ids.cpu() % 4096]

    def __call__(self, input_ids=None, **kw):
        class O(tuple):
            pass
        h = self._emb(input_ids, self.cx).to(self.device)
        o = O((h,))
        # like CLIP's pooler (hidden state at the first EOS position) this does not depend on padding: the mean of
        # the rows up to and including the first <eos>
        e = self._emb(input_ids, 768)
        ids = input_ids.cpu()
        n = (ids == 1).int().argmax(dim=1) + 1
        o.pooler_output = torch.stack([e[r, :int(n[r])].mean(dim=0) for r in range(e.shape[0])]).to(self.device)
        return o

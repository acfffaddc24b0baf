"""
Tokenizers.  Real checkpoints use their own HF tokenizer (reference v1/__init__.py:26-34:
AutoTokenizer.from_pretrained(base, model_max_length=2048, add_bos_token=False,
add_eos_token=True, pad_token="<pad>", padding_side="right", legacy=False)); no tokenizer files
exist offline, so synthetic-weight runs use SyntheticTokenizer: a deterministic id<->string map
with the attribute surface the inference code touches (infer/generate.py:212,234,240,287,383).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Union


class SyntheticTokenizer:
    """ids: pad, bos(image token), eos are special; the next 256 ids are the bytes; the rest are
    short TikZ-flavoured fragments, some of which contain newlines (so the MCTS line-splitting of
    infer/generate.py:229-272 is exercised)."""

    _FRAG = ["\\draw", " (", ") ", "--", "cycle", ";\n", "\\node", " at ", "{", "}", "[", "]",
             ",", "0.", "1", "2", "\\begin{tikzpicture}\n", "\\end{tikzpicture}\n", " circle",
             "rectangle ", "\\fill", "thick", "->", "\n", "%\n", "\\documentclass{standalone}\n"]

    def __init__(self, vocab_size: int, bos_token_id: int = 1, eos_token_id: int = 2,
                 pad_token_id: int = 0, model_max_length: int = 2048, image_token_id: int = None):
        self.vocab_size = vocab_size
        self.bos_token_id, self.eos_token_id, self.pad_token_id = bos_token_id, eos_token_id, pad_token_id
        self.bos_token, self.eos_token, self.pad_token = "<s>", "</s>", "<pad>"
        self.model_max_length = model_max_length
        self.padding_side = "right"
        self.init_kwargs: dict = {}
        special = {pad_token_id: self.pad_token, bos_token_id: self.bos_token, eos_token_id: self.eos_token}
        # v2: a dedicated image token (reference processing_detikzify.py:52 "<|reserved_special_token_2|>"); v1 reuses BOS
        self.image_token = "<|reserved_special_token_2|>"
        self.image_token_id = image_token_id if image_token_id not in special else None
        if self.image_token_id is not None:
            special[self.image_token_id] = self.image_token
        self.all_special_ids = sorted(special)
        self._id2tok: List[str] = []
        free = [i for i in range(vocab_size) if i not in special]
        table: Dict[int, str] = dict(special)
        for n, i in enumerate(free):
            if n < 256:
                table[i] = chr(n) if n < 128 else f"<0x{n:02X}>"
            else:
                f = self._FRAG[(n - 256) % len(self._FRAG)]
                table[i] = f if (n - 256) < len(self._FRAG) else f"{f}{n}"
        self._id2tok = [table[i] for i in range(vocab_size)]
        self.vocab: Dict[str, int] = {}
        for i, t in enumerate(self._id2tok):
            self.vocab.setdefault(t, i)
        self._byte_id = {chr(n): free[n] for n in range(min(128, len(free)))}

    def __len__(self):
        return self.vocab_size

    def get_vocab(self):
        return dict(self.vocab)

    def convert_ids_to_tokens(self, ids):
        if isinstance(ids, int):
            return self._id2tok[ids]
        return [self._id2tok[int(i)] for i in ids]

    def convert_tokens_to_ids(self, toks):
        if isinstance(toks, str):
            return self.vocab[toks]
        return [self.vocab[t] for t in toks]

    def decode(self, token_ids, skip_special_tokens: bool = False, **_):
        if hasattr(token_ids, "tolist"):
            token_ids = token_ids.tolist()
        if isinstance(token_ids, int):
            token_ids = [token_ids]
        out = []
        for i in token_ids:
            if skip_special_tokens and i in self.all_special_ids:
                continue
            out.append(self._id2tok[int(i)])
        return "".join(out)

    def batch_decode(self, seqs, **kw):
        return [self.decode(s, **kw) for s in seqs]

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        ids: List[int] = []
        i = 0
        specials = [(self.bos_token, self.bos_token_id), (self.eos_token, self.eos_token_id),
                    (self.pad_token, self.pad_token_id)]
        if self.image_token_id is not None:
            specials.append((self.image_token, self.image_token_id))
        while i < len(text):
            for s, sid in specials:
                if text.startswith(s, i):
                    ids.append(sid)
                    i += len(s)
                    break
            else:
                ch = text[i]
                ids.append(self._byte_id.get(ch, self._byte_id.get("?", 0)))
                i += 1
        return ids

    def __call__(self, text: Union[str, Iterable[str]], add_special_tokens: bool = False,
                 padding=False, truncation=False, max_length=None, return_tensors=None, **_):
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        enc = [self.encode(t, add_special_tokens) for t in texts]
        if truncation:
            lim = max_length or self.model_max_length
            enc = [e[:lim] for e in enc]
        mask = [[1] * len(e) for e in enc]
        return {"input_ids": enc, "attention_mask": mask}


def load_tokenizer(path: str, model_max_length: int = 2048, arch: str = "v1"):
    """HF tokenizer of a real checkpoint.  v1: configured as the reference does (v1/__init__.py:26-34: the base model's
    tokenizer gets a `<pad>` token, no BOS, EOS appended, length 2048).  v2: the checkpoint ships its own processor
    (`AutoProcessor.from_pretrained`, model/__init__.py:44): the tokenizer is taken as it was saved — overriding its pad
    token would append a new id behind the embedding table."""
    from transformers import AutoTokenizer
    if arch == "v2":
        return AutoTokenizer.from_pretrained(path)
    return AutoTokenizer.from_pretrained(path, model_max_length=model_max_length, add_bos_token=False,
                                         add_eos_token=True, pad_token="<pad>", padding_side="right",
                                         legacy=False)

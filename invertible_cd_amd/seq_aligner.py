"""Token-sequence alignment helpers behind the prompt-to-prompt controllers (API mirror of utils/seq_aligner.py).

`get_refinement_mapper` aligns the token sequences of a base prompt and an edited prompt with a global
(Needleman-Wunsch) alignment - gap 0, match +1, mismatch -1, ties resolved left > up > diagonal exactly as
utils/seq_aligner.py:48-64 does - and returns, per target token, the source token it maps to (-1 for inserted
tokens) plus the 0/1 "was aligned" weights.  `get_replacement_mapper` builds the 77x77 word-swap matrix for prompts of
equal word count (utils/seq_aligner.py:139-181).  CPU, once per controller.  Pinned by tests/golden/seq_aligner.npz.
"""
import numpy as np
import torch

LEFT, UP, DIAG, STOP = 1, 2, 3, 4


class ScoreParams:
    def __init__(self, gap, match, mismatch):
        self.gap, self.match, self.mismatch = gap, match, mismatch

    def mis_match_char(self, x, y):
        return self.match if x == y else self.mismatch


def global_align(x, y, score):
    """Returns (score matrix, trace matrix) of the global alignment of sequences x (rows) and y (columns)."""
    nx, ny = len(x), len(y)
    S = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    T = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    S[0, 1:] = (np.arange(ny) + 1) * score.gap
    S[1:, 0] = (np.arange(nx) + 1) * score.gap
    T[0, 1:], T[1:, 0], T[0, 0] = LEFT, UP, STOP
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            left = S[i, j - 1] + score.gap
            up = S[i - 1, j] + score.gap
            diag = S[i - 1, j - 1] + score.mis_match_char(x[i - 1], y[j - 1])
            best = max(left, up, diag)
            S[i, j] = best
            T[i, j] = LEFT if best == left else (UP if best == up else DIAG)
    return S, T


def get_aligned_sequences(x, y, trace_back):
    """Walk the trace from the bottom-right corner; returns (x aligned, y aligned, mapper[(j, i or -1)])."""
    xs, ys, pairs = [], [], []
    i, j = len(x), len(y)
    while i > 0 or j > 0:
        move = trace_back[i, j]
        if move == DIAG:
            i, j = i - 1, j - 1
            xs.append(x[i]); ys.append(y[j]); pairs.append((j, i))
        elif move == LEFT:
            j -= 1
            xs.append("-"); ys.append(y[j]); pairs.append((j, -1))
        elif move == UP:
            i -= 1
            xs.append(x[i]); ys.append("-")
        else:
            break
    pairs.reverse()
    return xs, ys, torch.tensor(pairs, dtype=torch.int64)


def get_mapper(x: str, y: str, tokenizer, max_len=77):
    xt, yt = tokenizer.encode(x), tokenizer.encode(y)
    _, trace = global_align(xt, yt, ScoreParams(0, 1, -1))
    base = get_aligned_sequences(xt, yt, trace)[-1]
    n = base.shape[0]
    alphas = torch.ones(max_len)
    alphas[:n] = base[:, 1].ne(-1).float()
    mapper = torch.zeros(max_len, dtype=torch.int64)
    mapper[:n] = base[:, 1]
    mapper[n:] = len(yt) + torch.arange(max_len - len(yt))
    return mapper, alphas


def get_refinement_mapper(prompts, tokenizer, max_len=77):
    pairs = [get_mapper(prompts[0], p, tokenizer, max_len) for p in prompts[1:]]
    return torch.stack([m for m, _ in pairs]), torch.stack([a for _, a in pairs])


def get_word_inds(text: str, word_place, tokenizer):
    words = text.split(" ")
    if type(word_place) is str:
        word_place = [i for i, w in enumerate(words) if w == word_place]
    elif type(word_place) is int:
        word_place = [word_place]
    found = []
    if len(word_place) > 0:
        pieces = [tokenizer.decode([tok]).strip("#") for tok in tokenizer.encode(text)][1:-1]
        consumed, wi = 0, 0
        for ti, piece in enumerate(pieces):
            consumed += len(piece)
            if wi in word_place:
                found.append(ti + 1)
            if consumed >= len(words[wi]):
                wi += 1
                consumed = 0
    return np.array(found)


def get_replacement_mapper_(x: str, y: str, tokenizer, max_len=77):
    wx, wy = x.split(" "), y.split(" ")
    if len(wx) != len(wy):
        raise ValueError(f"attention replacement edit can only be applied on prompts with the same length"
                         f" but prompt A has {len(wx)} words and prompt B has {len(wy)} words.")
    changed = [k for k in range(len(wy)) if wy[k] != wx[k]]
    src = [get_word_inds(x, k, tokenizer) for k in changed]
    dst = [get_word_inds(y, k, tokenizer) for k in changed]
    M = np.zeros((max_len, max_len))
    i = j = cur = 0
    while i < max_len and j < max_len:
        if cur < len(src) and src[cur][0] == i:
            s, d = src[cur], dst[cur]
            if len(s) == len(d):
                M[s, d] = 1
            else:
                for t in d:
                    M[s, t] = 1 / len(d)
            cur += 1
            i += len(s)
            j += len(d)
        elif cur < len(src):
            M[i, j] = 1
            i += 1
            j += 1
        else:
            M[j, j] = 1
            i += 1
            j += 1
    return torch.from_numpy(M).float()


def get_replacement_mapper(prompts, tokenizer, max_len=77):
    return torch.stack([get_replacement_mapper_(prompts[0], p, tokenizer, max_len) for p in prompts[1:]])

"""Token-sequence alignment behind the prompt-to-prompt edit controllers (same entry points as utils/seq_aligner.py).

    get_refinement_mapper(prompts, tokenizer)  -> (mapper int64 [P-1, 77], alphas fp32 [P-1, 77])      utils/seq_aligner.py:118-126
    get_replacement_mapper(prompts, tokenizer) -> fp32 [P-1, 77, 77]                                   utils/seq_aligner.py:184-191
    get_word_inds(text, word_place, tokenizer) -> token positions of a word                           utils/seq_aligner.py:129-151

What the reference computes, restated:
  * refinement: a global alignment of the two token-id sequences with gap 0 / match +1 / mismatch -1, predecessor priority
    left > up > diagonal on ties (utils/seq_aligner.py:48-64), then per target token the aligned source token or -1.
  * replacement: prompts with equal word counts; every changed word's source tokens map to its target tokens (1:1 when the
    token counts agree, uniformly 1/len(target) otherwise), unchanged tokens map one to one with the running offset, and the
    tail after the last changed word is the identity on the TARGET index (utils/seq_aligner.py:154-181).

How it is built here (own design, results pinned bit for bit by tests/golden/seq_aligner.npz, which was captured from the
reference): the alignment table is filled as an anti-diagonal wavefront of whole-vector torch ops - cell (i, j) only
depends on diagonals i+j-1 and i+j-2, so each of the nx+ny diagonals is one vectorised max over three shifted slices and the
table can live on the GPU (`device=`); the walk back is a <= nx+ny step loop over a single host copy of the move table.  The
replacement matrix is assembled from word -> token spans with slice assignments (no token-by-token state machine).  Both run
once per controller construction, never inside the U-Net's attention hooks.
"""
import numpy as np
import torch

LEFT, UP, DIAG, STOP = 1, 2, 3, 4          # move codes of the trace table (values as in the reference)


class ScoreParams:
    """(gap, match, mismatch) of the alignment - kept for callers of the reference API."""

    def __init__(self, gap, match, mismatch):
        self.gap, self.match, self.mismatch = gap, match, mismatch

    def mis_match_char(self, x, y):
        return self.match if x == y else self.mismatch


# ------------------------------------------------------------------------------------------------ alignment
def alignment_tables(x_ids, y_ids, gap=0, match=1, mismatch=-1, device="cpu"):
    """Score and move tables [nx+1, ny+1] of the global alignment of x (rows) and y (columns), wavefront order.

    Diagonal k holds the cells i + j == k.  With S stored row-major, the three predecessors of the cells of diagonal k are
    slices of diagonals k-1 (left: (i, j-1); up: (i-1, j)) and k-2 (diag: (i-1, j-1)); ties pick left, then up, then diag."""
    x = torch.as_tensor(list(x_ids), dtype=torch.int64, device=device)
    y = torch.as_tensor(list(y_ids), dtype=torch.int64, device=device)
    nx, ny = x.numel(), y.numel()
    S = torch.zeros((nx + 1, ny + 1), dtype=torch.int32, device=device)
    T = torch.full((nx + 1, ny + 1), STOP, dtype=torch.int32, device=device)
    S[0, 1:] = torch.arange(1, ny + 1, device=device, dtype=torch.int32) * gap
    S[1:, 0] = torch.arange(1, nx + 1, device=device, dtype=torch.int32) * gap
    T[0, 1:], T[1:, 0] = LEFT, UP
    sub = torch.where(x[:, None] == y[None, :], match, mismatch).to(torch.int32)         # substitution scores [nx, ny]
    for k in range(2, nx + ny + 1):
        i = torch.arange(max(1, k - ny), min(nx, k - 1) + 1, device=device)              # interior cells of the diagonal
        j = k - i
        left = S[i, j - 1] + gap
        up = S[i - 1, j] + gap
        diag = S[i - 1, j - 1] + sub[i - 1, j - 1]
        best = torch.maximum(torch.maximum(left, up), diag)
        S[i, j] = best
        T[i, j] = torch.where(best == left, LEFT, torch.where(best == up, UP, DIAG)).to(torch.int32)
    return S, T


def global_align(x, y, score):
    """Reference-shaped wrapper: (score matrix, trace matrix) as numpy int32 arrays."""
    S, T = alignment_tables(x, y, score.gap, score.match, score.mismatch)
    return S.numpy(), T.numpy()


def target_to_source(trace, nx, ny):
    """Walk the move table back from (nx, ny): for every target position j (in order) the aligned source position or -1."""
    trace = np.asarray(trace)
    i, j = nx, ny
    src_of = []
    while (i > 0 or j > 0) and trace[i, j] != STOP:
        move = trace[i, j]
        if move == DIAG:
            i, j = i - 1, j - 1
            src_of.append((j, i))
        elif move == LEFT:
            j -= 1
            src_of.append((j, -1))
        else:                                      # UP: a source token with no counterpart
            i -= 1
    return src_of[::-1]


def get_aligned_sequences(x, y, trace_back):
    """Reference-shaped: (x aligned, y aligned, mapper tensor [(target j, source i or -1)])."""
    pairs = target_to_source(trace_back, len(x), len(y))
    xs = [x[i] if i >= 0 else "-" for _, i in pairs]
    ys = [y[j] for j, _ in pairs]
    return xs, ys, torch.tensor(pairs, dtype=torch.int64).reshape(-1, 2)


def get_mapper(x: str, y: str, tokenizer, max_len=77, device="cpu"):
    """(mapper, alphas) of one (base, edit) prompt pair: mapper[j] = source token of target token j (-1: inserted),
    alphas[j] = 1 where a source token exists; positions past the target sequence continue the identity."""
    xt, yt = tokenizer.encode(x), tokenizer.encode(y)
    _, T = alignment_tables(xt, yt, device=device)
    pairs = target_to_source(T.cpu().numpy(), len(xt), len(yt))
    src = torch.tensor([i for _, i in pairs], dtype=torch.int64)
    n = src.numel()
    mapper = torch.cat([src, len(yt) + torch.arange(max_len - n)])[:max_len]
    alphas = torch.ones(max_len)
    alphas[:n] = (src >= 0).float()
    return mapper, alphas


def get_refinement_mapper(prompts, tokenizer, max_len=77):
    both = [get_mapper(prompts[0], p, tokenizer, max_len) for p in prompts[1:]]
    return torch.stack([m for m, _ in both]), torch.stack([a for _, a in both])


# ------------------------------------------------------------------------------------------------ words <-> tokens
def token_words(text: str, tokenizer):
    """For each token between BOS and EOS the index of the space-separated word it belongs to: a word owns consecutive
    token pieces until their characters cover it (the rule of utils/seq_aligner.py:139-150)."""
    words = text.split(" ")
    pieces = [tokenizer.decode([tok]).strip("#") for tok in tokenizer.encode(text)][1:-1]
    owner, w, covered = [], 0, 0
    for piece in pieces:
        owner.append(w)
        covered += len(piece)
        if w < len(words) and covered >= len(words[w]):
            w, covered = w + 1, 0
    return np.asarray(owner, dtype=np.int64), words


def get_word_inds(text: str, word_place, tokenizer):
    """Token positions (1-based: position 0 is BOS) of a word given by value (every occurrence) or by word index."""
    owner, words = token_words(text, tokenizer)
    if isinstance(word_place, str):
        wanted = [k for k, w in enumerate(words) if w == word_place]
    elif isinstance(word_place, int):
        wanted = [word_place]
    else:
        wanted = list(word_place)
    if len(wanted) == 0:
        return np.array([])
    return np.flatnonzero(np.isin(owner, wanted)) + 1


def get_replacement_mapper_(x: str, y: str, tokenizer, max_len=77):
    wx, wy = x.split(" "), y.split(" ")
    if len(wx) != len(wy):
        raise ValueError(f"attention replacement edit can only be applied on prompts with the same length"
                         f" but prompt A has {len(wx)} words and prompt B has {len(wy)} words.")
    own_x, _ = token_words(x, tokenizer)
    own_y, _ = token_words(y, tokenizer)
    M = np.zeros((max_len, max_len))
    i = j = 0                                              # next unmapped source / target token position
    for k in (k for k in range(len(wy)) if wx[k] != wy[k]):
        s, d = np.flatnonzero(own_x == k) + 1, np.flatnonzero(own_y == k) + 1
        if s.size == 0 or s[0] < i:
            continue
        run = min(int(s[0]) - i, max_len - i, max_len - j)   # unchanged tokens before the word: one to one, offset i - j
        r = np.arange(run)
        M[i + r, j + r] = 1
        i, j = i + run, j + run
        if i >= max_len or j >= max_len:
            break
        if s.size == d.size:
            M[s, d] = 1
        else:
            M[np.ix_(s, d)] = 1.0 / d.size
        i, j = i + s.size, j + d.size
    else:
        tail = np.arange(j, max(j, min(max_len, max_len - (i - j))))     # after the last edit: identity on the target index
        M[tail, tail] = 1
    return torch.from_numpy(M).float()


def get_replacement_mapper(prompts, tokenizer, max_len=77):
    return torch.stack([get_replacement_mapper_(prompts[0], p, tokenizer, max_len) for p in prompts[1:]])

"""Independent float64 NumPy restatement of the Gumbel-MuZero search (public mctx algorithm) with the
epistemic fields of emctx -- written array-style like the JAX original, one tree at a time.  It exists to cross-check
the C oracle (different language, different precision, different code structure); small cases only."""
import math

import numpy as np

F32_MIN = float(np.finfo(np.float32).min)
TINY = float(np.finfo(np.float32).tiny)


def considered_visits_sequence(m, S):
    if m <= 1:
        return list(range(S))
    log2max = int(math.ceil(math.log2(m)))
    seq, visits, nc = [], [0] * m, m
    while len(seq) < S:
        extra = max(1, int(S / (log2max * nc)))
        for _ in range(extra):
            seq.extend(visits[:nc])
            for i in range(nc):
                visits[i] += 1
        nc = max(2, nc // 2)
    return seq[:S]


def softmax(x):
    e = np.exp(x - x.max())
    return e / e.sum()


def mask_invalid(logits, invalid):
    return np.where(invalid, F32_MIN, logits - logits.max())


class Tree:
    def __init__(self, S, A, opts):
        N = S + 1
        self.S, self.N, self.A, self.o = S, N, A, opts
        self.node_visits = np.zeros(N, np.int64)
        self.raw_values = np.zeros(N); self.node_values = np.zeros(N)
        self.raw_var = np.zeros(N); self.node_var = np.zeros(N)
        self.parents = np.full(N, -1); self.action_from_parent = np.full(N, -1)
        self.children_index = np.full((N, A), -1)
        self.children_prior_logits = np.zeros((N, A)); self.children_values = np.zeros((N, A))
        self.children_visits = np.zeros((N, A), np.int64); self.children_rewards = np.zeros((N, A))
        self.children_discounts = np.zeros((N, A)); self.children_var = np.zeros((N, A)); self.children_rvar = np.zeros((N, A))
        self.embeddings = [None] * N

    def qvalues(self, n):
        return self.children_rewards[n] + self.children_discounts[n] * self.children_values[n]

    def qvar(self, n):
        return self.children_rvar[n] + self.children_discounts[n] ** 2 * self.children_var[n]

    def mixed(self, raw, q, n, probs):
        visits = self.children_visits[n]
        sumN = visits.sum()
        p = np.maximum(TINY, probs)
        sum_probs = np.where(visits > 0, p, 0.0).sum()
        wq = np.where(visits > 0, p * q / np.where(visits > 0, sum_probs, 1.0), 0.0).sum()
        return (raw + sumN * wq) / (sumN + 1)

    def qtransform(self, n):
        o = self.o
        visits = self.children_visits[n]
        probs = softmax(self.children_prior_logits[n])
        q, qv = self.qvalues(n), self.qvar(n)
        value = self.mixed(self.raw_values[n], q, n, probs)
        vvar = self.mixed(self.raw_var[n], qv, n, probs)
        bonus = o["beta"] * np.sqrt(np.where(visits > 0, qv, vvar))
        cq = np.where(visits > 0, q, value)
        if not o.get("bonus_after_rescale"):
            cq = cq + bonus
        if o["rescale"]:
            cq = (cq - cq.min()) / max(cq.max() - cq.min(), 1e-8)
        if o.get("bonus_after_rescale"):
            cq = cq + bonus
        return (50.0 + visits.max()) * 0.1 * cq

    def score_considered(self, cv, cq):
        logits = self.children_prior_logits[0]
        logits = logits - logits.max()
        pen = np.where(self.children_visits[0] == cv, 0.0, -np.inf)
        return np.maximum(-1e9, self.o["gumbel"] + logits + cq) + pen


def search(S, A, root_logits, root_value, root_var, root_state, recurrent_fn, gumbel, invalid, beta, discount, rescale=True,
           max_considered=16, bonus_after_rescale=False, var_backup=0):
    """recurrent_fn(state, action) -> (reward, discount, masked_logits, value, var, next_state)
    bonus_after_rescale / var_backup: the two unpinned epistemic choices (include/eaz.h eaz_search_opts)."""
    opts = dict(beta=beta, rescale=rescale, gumbel=np.asarray(gumbel, np.float64), bonus_after_rescale=bonus_after_rescale)
    t = Tree(S, A, opts)
    invalid = np.asarray(invalid, bool)
    t.children_prior_logits[0] = mask_invalid(np.asarray(root_logits, np.float64), invalid)
    t.raw_values[0] = t.node_values[0] = root_value
    t.raw_var[0] = t.node_var[0] = root_var
    t.node_visits[0] = 1
    t.embeddings[0] = root_state
    table = [considered_visits_sequence(m, S) for m in range(max_considered + 1)]
    for sim in range(S):
        node, depth = 0, 0
        while True:
            cq = t.qtransform(node)
            if depth == 0:
                nc = min(max_considered, int((~invalid).sum()))
                cv = table[nc][int(t.children_visits[0].sum())]
                a = int(np.argmax(np.where(invalid, -np.inf, t.score_considered(cv, cq))))
            else:
                visits = t.children_visits[node]
                p = softmax(t.children_prior_logits[node] + cq)
                a = int(np.argmax(p - visits / (1 + visits.sum())))
            nxt = t.children_index[node, a]
            depth += 1
            if nxt == -1 or depth >= S:
                break
            node = nxt
        parent = node
        if nxt == -1:
            nxt = sim + 1
        r, d, lg, v, var, ns = recurrent_fn(t.embeddings[parent], a)
        t.children_prior_logits[nxt] = lg
        t.raw_values[nxt] = t.node_values[nxt] = v
        t.raw_var[nxt] = t.node_var[nxt] = var
        t.node_visits[nxt] += 1
        t.embeddings[nxt] = ns
        t.children_index[parent, a] = nxt
        t.children_rewards[parent, a] = r
        t.children_discounts[parent, a] = d
        t.parents[nxt], t.action_from_parent[nxt] = parent, a
        G, U, idx = t.node_values[nxt], t.node_var[nxt], nxt
        while idx != 0:
            p_, a_ = t.parents[idx], t.action_from_parent[idx]
            cnt = t.node_visits[p_]
            G = t.children_rewards[p_, a_] + t.children_discounts[p_, a_] * G
            U = t.children_rvar[p_, a_] + t.children_discounts[p_, a_] ** 2 * U
            t.node_values[p_] = (t.node_values[p_] * cnt + G) / (cnt + 1.0)
            if var_backup == 0:
                t.node_var[p_] = (t.node_var[p_] * cnt + U) / (cnt + 1.0)          # running mean of the backed-up variances
            else:
                t.node_var[p_] = (t.node_var[p_] * cnt * cnt + U) / (cnt + 1.0) ** 2  # variance of the running mean
            t.node_visits[p_] = cnt + 1
            t.children_values[p_, a_] = t.node_values[idx]
            t.children_var[p_, a_] = t.node_var[idx]
            t.children_visits[p_, a_] += 1
            idx = p_
    cq = t.qtransform(0)
    action = int(np.argmax(np.where(invalid, -np.inf, t.score_considered(t.children_visits[0].max(), cq))))
    weights = softmax(mask_invalid(t.children_prior_logits[0] + cq, invalid))
    return t, action, weights

"""Thin object layer over libeaz.so: device context, environments, FC network, search, self-play.

PyTorch (ROCm) is used for device memory and streams only; every computation below is a HIP kernel of libeaz.so
reached through the C ABI of include/eaz.h.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import EazError, ReanalyzeCfg, SearchOpts, SelfplayCfg, check
from .params import fc_weight_list, resnet_array_list, HASH_MODULE, RESNET_HASH_MODULE, SIMHASH_MODULE, RESNET_SIMHASH_MODULE

_contexts: dict[int, "Context"] = {}

TREE_FIELDS = {
    "node_visits": (0, torch.int32), "raw_values": (1, torch.float32), "node_values": (2, torch.float32),
    "parents": (3, torch.int32), "action_from_parent": (4, torch.int32), "children_index": (5, torch.int32),
    "children_prior_logits": (6, torch.float32), "children_values": (7, torch.float32),
    "children_visits": (8, torch.int32), "children_rewards": (9, torch.float32),
    "children_discounts": (10, torch.float32), "raw_values_epistemic_variance": (11, torch.float32),
    "node_values_epistemic_variance": (12, torch.float32), "children_values_epistemic_variance": (13, torch.float32),
    "children_rewards_epistemic_variance": (14, torch.float32), "embeddings": (15, torch.uint8),
    "root_gumbel": (16, torch.float32), "beta": (17, torch.float32),
}


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libeaz needs contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def as_key(key) -> int:
    """Keys are 64-bit integers; (k0, k1) pairs and 2-element arrays are accepted."""
    if isinstance(key, (tuple, list, np.ndarray)) or (isinstance(key, torch.Tensor) and key.numel() == 2):
        k = [int(x) for x in (key.tolist() if hasattr(key, "tolist") else key)]
        return (k[0] & 0xFFFFFFFF) | ((k[1] & 0xFFFFFFFF) << 32)
    return int(key) & 0xFFFFFFFFFFFFFFFF


def key_pair(key) -> tuple[int, int]:
    k = as_key(key)
    return k & 0xFFFFFFFF, k >> 32


def fold(key, tag: int) -> int:
    out = C.c_uint64()
    check(_lib.load().eaz_key_fold(as_key(key), tag & 0xFFFFFFFF, C.byref(out)))
    return int(out.value)


def fingerprint(arrays) -> bytes:
    """Content fingerprint of a list of numpy arrays (xxh3-128 of the raw bytes, shapes and dtypes): the staleness test of every
    host-side weight / state copy that mirrors a pytree the caller may mutate in place."""
    import xxhash

    h = xxhash.xxh3_128()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(repr((a.shape, a.dtype.str)).encode())
        h.update(a.view(np.uint8).reshape(-1))
    return h.digest()


def tune(fc_tile: Optional[int] = None, fused_leaf: Optional[int] = None, tree_block: Optional[int] = None,
         tree_spread: Optional[int] = None, persist: Optional[int] = None, shards: Optional[int] = None,
         subleq_wave: Optional[int] = None, lazy_rows: Optional[int] = None) -> None:
    """Kernel-selection knobs (include/eaz.h eaz_tune_set); results are identical for every setting."""
    for knob, v in ((0, fc_tile), (1, fused_leaf), (2, tree_block), (3, tree_spread), (4, persist), (5, shards), (6, subleq_wave),
                    (7, lazy_rows)):
        if v is not None:
            check(_lib.load().eaz_tune_set(knob, int(v)))


class Context:
    """One per device (eaz_ctx)."""

    def __init__(self, device: int):
        if not torch.cuda.is_available():
            raise EazError("no ROCm device visible: the engine has no CPU fallback")
        self.device = device
        self.torch_device = torch.device("cuda", device)
        h = C.c_void_p()
        check(_lib.load().eaz_ctx_create(device, C.byref(h)))
        self.h = h

    @staticmethod
    def get(device: Optional[int] = None) -> "Context":
        if not torch.cuda.is_available():
            raise EazError("no ROCm device visible: the engine has no CPU fallback")
        device = torch.cuda.current_device() if device is None else device
        if device not in _contexts:
            _contexts[device] = Context(device)
        return _contexts[device]


SUBLEQ_REWARD_FNS = ("solved_or_not", "lowest_bytes")  # include/eaz.h EAZ_SUBLEQ_REWARD_*


class EnvHandle:
    """eaz_env + batched operations on compact state blobs (uint8 [B, state_bytes] device tensors)."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self.h = ctx, handle
        v = [C.c_int() for _ in range(5)]
        check(_lib.load().eaz_env_info(handle, *[C.byref(x) for x in v]))
        self.kind, self.num_actions, rows, cols, self.state_bytes = [x.value for x in v]
        self.obs_shape = (rows, cols) if self.kind != 2 else (6, 7, 2)  # connect_four observations are [6,7,2]
        self.obs_dim = rows * cols
        self.num_players = _lib.load().eaz_env_num_players(handle)

    @classmethod
    def deepsea(cls, size_of_grid: int, action_map: Optional[np.ndarray] = None, device=None):
        ctx = Context.get(device)
        am = None if action_map is None else np.ascontiguousarray(action_map, np.uint8).reshape(-1)
        assert am is None or am.size == size_of_grid * size_of_grid
        h = C.c_void_p()
        check(_lib.load().eaz_env_create_deepsea(ctx.h, size_of_grid, None if am is None else am.ctypes.data_as(C.c_void_p), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def subleq(cls, word_size: int, tasks: Sequence[int], use_binary_encoding: bool, device=None, reward_fn: str = "solved_or_not"):
        """reward_fn: "solved_or_not" (subleq.py:535-537, the default of subleq.py:603) or "lowest_bytes" (:540-542)."""
        ctx = Context.get(device)
        t = np.ascontiguousarray(tasks, np.int32)
        if reward_fn not in SUBLEQ_REWARD_FNS:
            raise EazError(f"unknown Subleq reward function {reward_fn!r} (available: {', '.join(SUBLEQ_REWARD_FNS)})")
        h = C.c_void_p()
        check(_lib.load().eaz_env_create_subleq(ctx.h, word_size, t.ctypes.data_as(C.c_void_p), len(t), int(use_binary_encoding), C.byref(h)))
        env = cls(ctx, h)
        check(_lib.load().eaz_env_set_subleq_reward(h, SUBLEQ_REWARD_FNS.index(reward_fn)))
        return env

    @classmethod
    def connect4(cls, device=None):
        ctx = Context.get(device)
        h = C.c_void_p()
        check(_lib.load().eaz_env_create_connect4(ctx.h, C.byref(h)))
        return cls(ctx, h)

    def new_states(self, *lead) -> torch.Tensor:
        return torch.zeros(*lead, self.state_bytes, dtype=torch.uint8, device=self.ctx.torch_device)

    def init(self, B: int, key=0) -> torch.Tensor:
        s = self.new_states(B)
        check(_lib.load().eaz_env_init(self.h, B, as_key(key), _ptr(s), _stream()))
        return s

    def step(self, states: torch.Tensor, action: torch.Tensor, key=0, autoreset: bool = False) -> torch.Tensor:
        B = states.shape[0]
        out = torch.empty_like(states)
        a = action.to(torch.int32).contiguous()
        check(_lib.load().eaz_env_step(self.h, B, _ptr(states), _ptr(a), as_key(key), int(autoreset), _ptr(out), _stream()))
        return out

    def observe(self, states: torch.Tensor) -> torch.Tensor:
        flat = states.reshape(-1, self.state_bytes)
        obs = torch.empty(flat.shape[0], self.obs_dim, dtype=torch.uint8, device=states.device)
        check(_lib.load().eaz_env_observe(self.h, flat.shape[0], _ptr(flat), _ptr(obs), _stream()))
        return obs.reshape(*states.shape[:-1], self.obs_dim)

    def fields(self, states: torch.Tensor) -> dict:
        flat = states.reshape(-1, self.state_bytes)
        n, dev = flat.shape[0], states.device
        term = torch.empty(n, dtype=torch.uint8, device=dev)
        trunc = torch.empty(n, dtype=torch.uint8, device=dev)
        rew = torch.empty(n, self.num_players, dtype=torch.float32, device=dev)
        sc = torch.empty(n, dtype=torch.int32, device=dev)
        legal = torch.empty(n, self.num_actions, dtype=torch.uint8, device=dev)
        cp = torch.empty(n, dtype=torch.int32, device=dev)
        check(_lib.load().eaz_env_fields(self.h, n, _ptr(flat), _ptr(term), _ptr(trunc), _ptr(rew), _ptr(sc), _ptr(legal), _ptr(cp), _stream()))
        lead = states.shape[:-1]
        return dict(terminated=term.bool().reshape(lead), truncated=trunc.bool().reshape(lead),
                    rewards=rew.reshape(*lead, self.num_players), step_count=sc.reshape(lead),
                    legal_action_mask=legal.bool().reshape(*lead, self.num_actions), current_player=cp.reshape(lead))

    def random_play(self, B: int, T: int, key=0, want_traj: bool = False):
        states = self.new_states(B)
        traj = self.new_states(T, B) if want_traj else None
        check(_lib.load().eaz_random_play(self.h, B, T, as_key(key), _ptr(states), _ptr(traj), _stream()))
        return states, traj


class FCNet:
    """eaz_net for EpistemicFullyConnectedAZNet (src/network/fully_connected.py)."""

    hash_module = HASH_MODULE  # haiku module path of the novelty set inside model_state

    def __init__(self, ctx: Context, in_dim: int, hidden: int, num_actions: int, params, state=None, max_ube: float = 1.0,
                 max_epistemic_variance_reward: float = 1.0, hash_off: int = 0, hash_len: Optional[int] = None, hash_bits: int = 24,
                 input_order=None):
        self.ctx, self.in_dim, self.hidden, self.num_actions = ctx, in_dim, hidden, num_actions
        self.input_order = None
        self.hash_bits = hash_bits
        hash_len = in_dim - hash_off if hash_len is None else hash_len
        wl = fc_weight_list(params)
        self._keep = wl
        wp = (C.c_void_p * 12)(*[w.ctypes.data for w, _ in wl])
        bp = (C.c_void_p * 12)(*[b.ctypes.data for _, b in wl])
        h = C.c_void_p()
        check(_lib.load().eaz_net_create_fc(ctx.h, in_dim, hidden, num_actions, wp, bp, max_ube, max_epistemic_variance_reward,
                                            hash_off, hash_len, hash_bits, C.byref(h)))
        self.h = h
        self.params_fp = fingerprint([x for wb in wl for x in wb])
        self.state_fp = fingerprint([np.zeros(1 << (hash_bits - 3), np.uint8)])
        if state is not None:
            self.set_state(state)
        if input_order:
            self.set_input_order(input_order)

    def set_input_order(self, segments):
        """Layer-1 input order (include/eaz.h eaz_net_set_input_order; a switch of the numerics contract that the test oracle follows):
        `segments` = [(start, len), ...] whose concatenation is a permutation of range(in_dim), e.g. params.subleq_constants_first(16);
        None / [] restores the k-ascending default."""
        segments = [(int(a), int(b)) for a, b in (segments or [])]
        st = np.ascontiguousarray([a for a, _ in segments], np.int32)
        ln = np.ascontiguousarray([b for _, b in segments], np.int32)
        check(_lib.load().eaz_net_set_input_order(self.h, len(segments), st.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), _stream()))
        self.input_order = segments or None

    def set_params(self, params):
        wl = fc_weight_list(params)
        self._keep = wl
        wp = (C.c_void_p * 12)(*[w.ctypes.data for w, _ in wl])
        bp = (C.c_void_p * 12)(*[b.ctypes.data for _, b in wl])
        check(_lib.load().eaz_net_set_weights(self.h, wp, bp, _stream()))
        torch.cuda.current_stream().synchronize()
        self.params_fp = fingerprint([x for wb in wl for x in wb])

    def set_state(self, state):
        bs = np.ascontiguousarray(state[self.hash_module]["binary_set"], np.uint8)
        assert bs.size == 1 << (self.hash_bits - 3)
        check(_lib.load().eaz_net_set_bitset(self.h, bs.ctypes.data_as(C.c_void_p), _stream()))
        torch.cuda.current_stream().synchronize()
        self.state_fp = fingerprint([bs])

    def get_state(self):
        """haiku model_state with the device's current bit set (the only state the FC network has)."""
        bs = np.empty(1 << (self.hash_bits - 3), np.uint8)
        check(_lib.load().eaz_net_get_bitset(self.h, bs.ctypes.data_as(C.c_void_p), _stream()))
        self.state_fp = fingerprint([bs])
        mod = {"binary_set": bs}
        if getattr(self, "_sim_matrix", None) is not None:
            mod["random_matrix"] = self._sim_matrix
        return {self.hash_module: mod}

    def set_simhash(self, random_matrix, module: Optional[str] = None):
        """hash_class = SimHash (config.py:27, hashes.py:100-139): `random_matrix` is the haiku state '<net>/sim_hash/random_matrix'
        [hash_len, bits_per_hash].  The novelty set then lives under the sim_hash module name in model_state."""
        rm = np.ascontiguousarray(random_matrix, np.float32)
        self._sim_matrix = rm
        check(_lib.load().eaz_net_set_simhash(self.h, rm.ctypes.data_as(C.c_void_p), _stream()))
        self.hash_module = module or (RESNET_SIMHASH_MODULE if isinstance(self, ResNet) else SIMHASH_MODULE)

    def sync_model(self, params, state) -> bool:
        """Bring the device copy up to date with (params, state) if their CONTENT differs from what was uploaded last -- the caller may
        have replaced the pytrees or mutated them in place (the reference re-applies model_params on every call).  Returns True if
        anything was uploaded."""
        changed = False
        pfp = fingerprint(self._param_arrays(params, state))
        if pfp != self.params_fp:
            self.set_params(params) if not isinstance(self, ResNet) else self.set_params(params, state)
            changed = True
        if state is not None and self.hash_module in state:
            sfp = fingerprint([np.ascontiguousarray(state[self.hash_module]["binary_set"], np.uint8)])
            if sfp != self.state_fp:
                self.set_state(state)
                changed = True
        return changed

    def _param_arrays(self, params, state):
        return [x for wb in fc_weight_list(params) for x in wb]

    def forward(self, obs: torch.Tensor):
        """-> (exploitation_logits, exploration_logits, value, value_epistemic_variance, reward_epistemic_variance)"""
        o = obs.reshape(-1, self.in_dim)
        if o.dtype != torch.uint8:
            o = o.to(torch.uint8)
        o = o.contiguous()
        B, dev, A = o.shape[0], o.device, self.num_actions
        ex = torch.empty(B, A, dtype=torch.float32, device=dev)
        xp = torch.empty(B, A, dtype=torch.float32, device=dev)
        v = torch.empty(B, dtype=torch.float32, device=dev)
        u = torch.empty(B, dtype=torch.float32, device=dev)
        nov = torch.empty(B, dtype=torch.float32, device=dev)
        check(_lib.load().eaz_net_forward(self.h, B, _ptr(o), _ptr(ex), _ptr(xp), _ptr(v), _ptr(u), _ptr(nov), _stream()))
        return ex, xp, v, u, nov

    def hash_lookup(self, obs: torch.Tensor):
        o = obs.reshape(-1, self.in_dim).to(torch.uint8).contiguous()
        seen = torch.empty(o.shape[0], dtype=torch.uint8, device=o.device)
        idx = torch.empty(o.shape[0], dtype=torch.int32, device=o.device)
        check(_lib.load().eaz_hash_lookup(self.h, o.shape[0], _ptr(o), _ptr(seen), _ptr(idx), _stream()))
        return seen.bool(), idx

    def hash_update(self, obs: torch.Tensor):
        o = obs.reshape(-1, self.in_dim).to(torch.uint8).contiguous()
        check(_lib.load().eaz_hash_update(self.h, o.shape[0], _ptr(o), _stream()))


class ResNet(FCNet):
    """eaz_net for EpistemicResidualAZNet (src/network/resnet.py) on [H, W, Cin] observations; same methods as FCNet."""

    hash_module = RESNET_HASH_MODULE

    def __init__(self, ctx: Context, H: int, W: int, cin: int, num_actions: int, params, state, num_channels: int = 64,
                 num_blocks: int = 5, hash_bits: int = 24):
        self.ctx, self.in_dim, self.hidden, self.num_actions, self.hash_bits = ctx, H * W * cin, num_channels, num_actions, hash_bits
        self.num_blocks = num_blocks
        arrs = resnet_array_list(params, state, num_blocks)
        self._keep = arrs
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        h = C.c_void_p()
        check(_lib.load().eaz_net_create_resnet(ctx.h, H, W, cin, num_channels, num_blocks, num_actions, ptrs, len(arrs), hash_bits, C.byref(h)))
        self.h = h
        self.params_fp = fingerprint(arrs)
        self.state_fp = fingerprint([np.zeros(1 << (hash_bits - 3), np.uint8)])
        self._state = state  # BatchNorm statistics etc.: get_state() hands them back with the device's bit set
        if state is not None and RESNET_HASH_MODULE in state:
            self.set_state(state)

    def set_params(self, params, state=None):
        """New weights (and BatchNorm statistics, which live in `state`) into the SAME packed device buffer: searches and captured
        graphs built on this network keep working and see them (eaz_net_set_resnet_weights)."""
        state = self._state if state is None else state
        arrs = resnet_array_list(params, state, self.num_blocks)
        self._keep = arrs
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        check(_lib.load().eaz_net_set_resnet_weights(self.h, ptrs, len(arrs), _stream()))
        self.params_fp = fingerprint(arrs)
        self._state = state

    def get_state(self):
        """The model_state given last (BatchNorm mean / var EMAs, ...) with 'az_resnet/xxhash32'/binary_set replaced by the device's."""
        bs = np.empty(1 << (self.hash_bits - 3), np.uint8)
        check(_lib.load().eaz_net_get_bitset(self.h, bs.ctypes.data_as(C.c_void_p), _stream()))
        self.state_fp = fingerprint([bs])
        out = {k: dict(v) for k, v in (self._state or {}).items()}
        mod = {"binary_set": bs}
        if getattr(self, "_sim_matrix", None) is not None:
            mod["random_matrix"] = self._sim_matrix
        out[self.hash_module] = mod
        return out

    def _param_arrays(self, params, state):
        return resnet_array_list(params, self._state if state is None else state, self.num_blocks)


class Search:
    """eaz_search: Gumbel-MuZero search with epistemic fields over B trees of num_simulations + 1 nodes."""

    def __init__(self, env: EnvHandle, net: FCNet, B: int, num_simulations: int, discount: float = 0.997,
                 two_players_game: bool = False, exploration: bool = False, rescale_values: bool = True,
                 use_mixed_value: bool = True, value_scale: float = 0.1, maxvisit_init: float = 50.0,
                 max_num_considered_actions: int = 16, gumbel_scale: float = 1.0, max_depth: Optional[int] = None,
                 epistemic_bonus_after_rescale: bool = False, epistemic_variance_backup: int = 0):
        self.env, self.net, self.B, self.S, self.N, self.A = env, net, B, num_simulations, num_simulations + 1, env.num_actions
        self.opts = SearchOpts(num_simulations, max_depth or 0, max_num_considered_actions, gumbel_scale, int(rescale_values),
                               int(use_mixed_value), value_scale, maxvisit_init, discount, int(two_players_game), int(exploration),
                               int(epistemic_bonus_after_rescale), int(epistemic_variance_backup))
        h = C.c_void_p()
        check(_lib.load().eaz_search_create(env.ctx.h, env.h, net.h, B, C.byref(self.opts), C.byref(h)))
        self.h = h
        self.device = env.ctx.torch_device

    def __del__(self):
        try:
            _lib.load().eaz_search_destroy(self.h)
        except Exception:
            pass

    # -- step-wise API ------------------------------------------------------------------------------------------
    def begin(self, prior_logits, value, value_var, root_states, beta, invalid, gumbel):
        f = lambda t: t.to(torch.float32).contiguous()
        self._keep = [f(prior_logits), f(value), f(value_var), root_states.contiguous(), f(beta),
                      None if invalid is None else invalid.to(torch.uint8).contiguous(), f(gumbel)]
        k = self._keep
        check(_lib.load().eaz_search_begin(self.h, _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(k[3]), _ptr(k[4]), _ptr(k[5]), _ptr(k[6]), _stream()))

    def simulate(self, sim: int):
        check(_lib.load().eaz_search_simulate(self.h, sim, _stream()))

    def run(self):
        check(_lib.load().eaz_search_run(self.h, _stream()))

    def debug_poison(self, byte_value: int = 0x7F, child_rows: bool = False):
        """Test hook (include/eaz.h eaz_search_debug_poison): fill the select scratch arrays -- or, `child_rows`, the eight [B,N,A] arrays of
        the tree -- with a byte pattern."""
        check(_lib.load().eaz_search_debug_poison(self.h, 1 if child_rows else 0, byte_value, _stream()))

    def launch_info(self) -> dict:
        """one_launch: selfplay() / run() are ONE kernel launch on LDS-resident trees for this search under the current tuning; lds_bytes of it."""
        one, lds = C.c_int(), C.c_int()
        check(_lib.load().eaz_search_launch_info(self.h, C.byref(one), C.byref(lds)))
        return dict(one_launch=bool(one.value), lds_bytes=int(lds.value))

    def finish(self):
        act = torch.empty(self.B, dtype=torch.int32, device=self.device)
        aw = torch.empty(self.B, self.A, dtype=torch.float32, device=self.device)
        check(_lib.load().eaz_search_finish(self.h, _ptr(act), _ptr(aw), _stream()))
        return act, aw

    def summary(self) -> dict:
        B, A, dev = self.B, self.A, self.device
        o = dict(value=torch.empty(B, device=dev), value_epistemic_std=torch.empty(B, device=dev),
                 qvalues=torch.empty(B, A, device=dev), qvalues_epistemic_variance=torch.empty(B, A, device=dev),
                 visit_counts=torch.empty(B, A, device=dev), visit_probs=torch.empty(B, A, device=dev))
        check(_lib.load().eaz_search_summary(self.h, _ptr(o["value"]), _ptr(o["value_epistemic_std"]), _ptr(o["qvalues"]),
                                             _ptr(o["qvalues_epistemic_variance"]), _ptr(o["visit_counts"]), _ptr(o["visit_probs"]), _stream()))
        return o

    def select(self, sim: int):
        p = torch.empty(self.B, dtype=torch.int32, device=self.device)
        a = torch.empty_like(p)
        n = torch.empty_like(p)
        check(_lib.load().eaz_search_select(self.h, sim, _ptr(p), _ptr(a), _ptr(n), _stream()))
        return p, a, n

    def expand(self, sim: int, reward, reward_var, discount, prior_logits, value, value_var):
        f = lambda t: t.to(torch.float32).contiguous()
        keep = [f(reward), f(reward_var), f(discount), f(prior_logits), f(value), f(value_var)]
        check(_lib.load().eaz_search_expand(self.h, sim, *[_ptr(t) for t in keep], _stream()))
        torch.cuda.current_stream().synchronize()  # keep the temporaries alive until the kernel has consumed them

    def field(self, name: str) -> torch.Tensor:
        """Copy of one tree array (mctx Tree / emctx EpistemicTree field names)."""
        fid, dt = TREE_FIELDS[name]
        ptr, shape, eb = C.c_void_p(), (C.c_int64 * 3)(), C.c_int()
        check(_lib.load().eaz_search_tree_view(self.h, fid, C.byref(ptr), shape, C.byref(eb)))
        dims = tuple(int(d) for d in shape if d > 0)
        typestr = {torch.int32: "<i4", torch.float32: "<f4", torch.uint8: "|u1"}[dt]

        class _View:  # zero-copy view of library-owned device memory
            __cuda_array_interface__ = {"shape": dims, "typestr": typestr, "data": (int(ptr.value), False), "version": 2}

        return torch.as_tensor(_View(), device=self.device).clone()

    def dump(self) -> dict:
        return {k: self.field(k).cpu().numpy() for k in TREE_FIELDS if k != "embeddings"}

    # -- fused self-play driver ---------------------------------------------------------------------------------
    def selfplay(self, states: torch.Tensor, T: int, key=0, gumbel: Optional[torch.Tensor] = None,
                 directed_exploration: bool = False, exploration_beta: float = 0.0, uniform_search_policy: bool = False,
                 sample_actions: bool = False, sample_from_improved_policy: bool = False, outputs: Optional[dict] = None,
                 use_graph: bool = True) -> dict:
        """src/selfplay.py:78-149.  `states` ([B, state_bytes]) is advanced in place; returns SelfplayOutput fields [T,B,..]."""
        B, A, dev = self.B, self.A, self.device
        assert states.shape == (B, self.env.state_bytes) and states.is_contiguous()
        want = dict(state=((T, B, self.env.state_bytes), torch.uint8), root_value=((T, B), torch.float32),
                    root_epistemic_std=((T, B), torch.float32), value_prediction=((T, B), torch.float32),
                    ube_prediction=((T, B), torch.float32), q_values_epistemic_variance=((T, B, A), torch.float32),
                    action=((T, B), torch.int32))
        if outputs is not None:
            # reused output buffers are written by the device with these exact extents
            for k, (shape, dt) in want.items():
                t = outputs.get(k)
                if t is None and k == "action":
                    continue
                if t is None or tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != dev:
                    raise EazError(f"selfplay(outputs=...): `{k}` must be a contiguous {dt} tensor of shape {shape} on {dev}")
        if outputs is None:
            outputs = dict(
                state=torch.empty(T, B, self.env.state_bytes, dtype=torch.uint8, device=dev),
                root_value=torch.empty(T, B, device=dev), root_epistemic_std=torch.empty(T, B, device=dev),
                value_prediction=torch.empty(T, B, device=dev), ube_prediction=torch.empty(T, B, device=dev),
                q_values_epistemic_variance=torch.empty(T, B, A, device=dev),
                action=torch.empty(T, B, dtype=torch.int32, device=dev),
            )
        cfg = SelfplayCfg(T, int(directed_exploration), exploration_beta, int(uniform_search_policy), int(sample_actions),
                          int(sample_from_improved_policy), int(use_graph))
        g = None if gumbel is None else gumbel.to(torch.float32).contiguous()
        if g is not None and tuple(g.shape) != (T, B, A):
            raise EazError(f"selfplay(gumbel=...): expected shape {(T, B, A)}, got {tuple(g.shape)}")
        o = outputs
        # The captured per-step graph is keyed on the address of the state buffer: the search advances its OWN buffer and copies in / out
        # (2 x B x state_bytes), so callers that pass a fresh tensor every call (api.selfplay clones last_states) replay the same graph.
        if getattr(self, "_states_buf", None) is None:
            self._states_buf = torch.empty(B, self.env.state_bytes, dtype=torch.uint8, device=dev)
        self._states_buf.copy_(states)
        check(_lib.load().eaz_selfplay_run(self.h, C.byref(cfg), as_key(key), _ptr(self._states_buf), _ptr(g), _ptr(o["state"]),
                                           _ptr(o["root_value"]), _ptr(o["root_epistemic_std"]), _ptr(o["value_prediction"]),
                                           _ptr(o["ube_prediction"]), _ptr(o["q_values_epistemic_variance"]), _ptr(o.get("action")),
                                           _stream()))
        states.copy_(self._states_buf)
        return outputs

    # -- "next" drivers --------------------------------------------------------------------------------------------
    def evaluate(self, states: torch.Tensor, exploitation_beta: float = 0.0, max_episode_length: int = 1000):
        """src/evaluate.py:13-57 from `states` (advanced in place) -> (per-episode returns [B], loop iterations)."""
        ret = torch.empty(self.B, device=self.device)
        it = C.c_int()
        check(_lib.load().eaz_evaluate_run(self.h, exploitation_beta, max_episode_length, _ptr(states), _ptr(ret), C.byref(it), _stream()))
        return ret, it.value

    def reanalyze(self, states: torch.Tensor, next_states: torch.Tensor, key=0, gumbel: Optional[torch.Tensor] = None,
                  reanalyze_beta: float = 0.0, exploration_beta: float = 0.0, exploration_ube_target: bool = True,
                  exploration_policy_target_temperature: float = 1.0) -> dict:
        """src/reanalyze.py:52-131 -> dict(value_target, ube_target, exploration_policy_target, exploitation_policy_target, action)."""
        B, A, dev = self.B, self.A, self.device
        o = dict(value_target=torch.empty(B, device=dev), ube_target=torch.empty(B, device=dev),
                 exploration_policy_target=torch.empty(B, A, device=dev), exploitation_policy_target=torch.empty(B, A, device=dev),
                 action=torch.empty(B, dtype=torch.int32, device=dev))
        cfg = ReanalyzeCfg(reanalyze_beta, exploration_beta, int(exploration_ube_target), exploration_policy_target_temperature)
        g = None if gumbel is None else gumbel.to(torch.float32).contiguous()
        check(_lib.load().eaz_reanalyze_run(self.h, C.byref(cfg), as_key(key), _ptr(states.contiguous()), _ptr(next_states.contiguous()), _ptr(g),
                                            _ptr(o["value_target"]), _ptr(o["ube_target"]), _ptr(o["exploration_policy_target"]),
                                            _ptr(o["exploitation_policy_target"]), _ptr(o["action"]), _stream()))
        return o

    # -- instrumentation ----------------------------------------------------------------------------------------
    def profile(self, on: bool):
        check(_lib.load().eaz_profile_enable(self.h, int(on)))

    def profile_replay(self, kernel_class: str, iters: int = 200) -> float:
        """Average duration (ms) of the last simulation's network forward / leaf gather, launched `iters` times back to back."""
        ms = C.c_double()
        check(_lib.load().eaz_profile_replay(self.h, _lib.K_NAMES.index(kernel_class), iters, C.byref(ms), _stream()))
        return ms.value / iters

    def profile_read(self, reset: bool = True) -> dict:
        ms = (C.c_double * len(_lib.K_NAMES))()
        n = (C.c_int64 * len(_lib.K_NAMES))()
        check(_lib.load().eaz_profile_read(self.h, ms, n, int(reset)))
        out = {name: dict(ms=ms[i], launches=int(n[i])) for i, name in enumerate(_lib.K_NAMES)}
        ks = C.c_uint64()
        check(_lib.load().eaz_profile_net_ksteps(self.h, C.byref(ks), int(reset)))
        out["net"]["layer1_ksteps"] = int(ks.value)
        return out

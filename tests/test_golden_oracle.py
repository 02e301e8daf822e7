"""The oracle reproduces the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import glob
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG  # noqa: E402

CASES = sorted(MG.CASES)


def test_all_cases_have_fixtures():
    assert sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz"))) == CASES


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    kind, size, B, S, T, seed, fill, flags = MG.CASES[name]
    z = np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    flags = dict(flags)
    env, net, discount = MG.build(kind, size, seed, fill, flags.pop("constants_first", False))
    st = env.init(B, key=(seed, 0))
    out = oracle.selfplay(env, net, st, T=T, S=S, key=(1000 + seed, 0), discount=discount, **flags)
    for k, v in out.items():
        assert np.array_equal(v, z[f"sp_{k}"]), k
    assert np.array_equal(st, z["last_states"])
    explore = bool(flags.get("directed_exploration", False))
    act, aw, summ, tree = oracle.search(env, net, st, S, z["root_logits"], z["root_value"], z["root_var"], z["beta"], z["gumbel"],
                                        discount=discount, use_exploration_logits=explore, two_player=bool(flags.get("two_player", False)))
    assert np.array_equal(act, z["search_action"]) and np.array_equal(aw, z["search_action_weights"])
    d = tree.dump()
    for k in MG.TREE_INT + MG.TREE_F32:
        assert np.array_equal(d[k], z[f"tree_{k}"]), k
    # domain sanity of the stored trajectories: rewards only at terminal DeepSea states, episodes reset after termination
    f = env.fields(z["sp_state"].reshape(T * B, -1))
    if kind == "deepsea":
        assert ((f["rewards"] > 0) <= f["terminated"]).all()
        assert (f["step_count"] <= size - 1).all()

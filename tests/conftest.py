import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _cpu_share():
    """Cores this process may use (affinity mask and cgroup quota): the oracle's OpenMP team is cut to it -- on a GPU box the process sees every
    hardware thread of the host but owns a 16-core quota, and a default-sized team's idle workers spin the quota away (bench.py oracle_module)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


@pytest.fixture(scope="session")
def oracle():
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # read by libgomp when it is loaded
    import eaz_oracle as O

    O.build()
    O.lib().eazo_set_num_threads(_cpu_share())
    return O


@pytest.fixture(scope="session")
def eaz():
    import e_alphazero_amd as m

    return m


# Every GPU test runs once per kernel selection (include/eaz.h eaz_tune_set): all of them must give identical results.
KERNEL_VARIANTS = {
    # name: (fc_tile, fused_leaf, tree_block, tree_spread, persist, shards, subleq_wave, lazy_rows)
    "persist": (0, 1, 64, 0, 1, 0, 1, 1),           # default: eligible self-play calls (DeepSea / Subleq-16, FC, trees fit LDS) run as ONE launch
    "persist-step": (0, 1, 64, 0, 2, 0, 1, 1),      # one launch per SEARCH: Subleq-16 subleq16_search_kernel + the per-real-step kernels around it
    "auto-fused": (0, 1, 64, 0, 0, 1, 1, 2),        # 16-row FC tiles, DeepSea leaf step fused into the tree kernel + one-hot forward; the per-step
                                                    # self-play driver never clears the [B,N,A] tree arrays (EAZ_TUNE_LAZY_ROWS = 2: every width)
    "tile16-unfused": (16, 0, 128, 4, 0, 1, 1, 0),  # 16-row FC tiles on dense observation rows, separate leaf-gather launch; cleared tree arrays
    "tile64-sharded": (64, 0, 256, 1, 0, 2, 1, 2),  # 64-row FC tiles, separate leaf-gather launch, 256-thread tree workgroups; every search of
                                                    # >= 128 trees runs as two half-batches on two streams (sub-batch pipelining); lazy rows
    "perlane-unfused": (16, 0, 128, 4, 0, 1, 0, 1), # Subleq word sizes other than 16 on the lane-per-environment kernel (subleq_step_kernel, the
                                                    # EAZ_TUNE_SUBLEQ_WAVE = 0 comparison path) -- generated only for PERLANE_TESTS
}
# the only tests whose kernels depend on EAZ_TUNE_SUBLEQ_WAVE (Subleq with a word size other than 16): the sixth variant runs just these
PERLANE_TESTS = ("test_subleq_word_sizes_that_are_not_multiples_of_four", "test_subleq_interpreter_on_random_programs")


def pytest_generate_tests(metafunc):
    if "kernel_variant" in metafunc.fixturenames:
        names = [v for v in KERNEL_VARIANTS if v != "perlane-unfused" or metafunc.function.__name__ in PERLANE_TESTS]
        metafunc.parametrize("kernel_variant", names, indirect=True)


@pytest.fixture
def kernel_variant(request):
    from e_alphazero_amd import engine

    tile, fused, tblock, tspread, persist, shards, wave, lazy = KERNEL_VARIANTS[request.param]
    engine.tune(fc_tile=tile, fused_leaf=fused, tree_block=tblock, tree_spread=tspread, persist=persist, shards=shards, subleq_wave=wave,
                lazy_rows=lazy)
    yield request.param
    engine.tune(fc_tile=0, fused_leaf=1, tree_block=64, tree_spread=0, persist=1, shards=0, subleq_wave=1, lazy_rows=1)

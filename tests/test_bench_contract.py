"""bench.py's command line and leg table, checked without a GPU: the driver runs `python bench.py --gpus N --steps K --warmup W` and the
line's legs are spelled as argument overrides -- a typo in one would only show on the GPU box."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_defaults_and_legs_are_well_formed():
    import bench
    a = bench.parse_defaults()
    assert a.gpus == 1 and a.steps >= 10 and a.warmup >= 1 and a.ndim == 100 and a.ntemps == 64 and a.nwalkers == 4096
    assert a.mix == "scam" and a.logl == "iso" and a.cov_mode == "pooled" and not a.callback
    names = [n for n, _, _, _ in bench.ALSO]
    assert len(names) == len(set(names)) and "config2_batched_callback" in names and "config4_share_1000d_64x512" in names
    parser = bench.make_parser()
    choices = {act.dest: act.choices for act in parser._actions if act.choices}
    for name, over, steps, warmup in bench.ALSO:
        b = copy.copy(a)
        for k, v in over.items():
            assert hasattr(b, k), (name, k)                              # every override is an argument of the parser
            if k in choices:
                assert v in choices[k], (name, k, v)
            setattr(b, k, v)
        assert steps >= 1 and warmup >= 1
        w = bench.cycle_weights(b)
        assert len(w) == 3 and sum(w) > 0
        if b.mix == "default" and not b.callback:
            assert warmup > 100                                          # DE joins the cycle after burn = 10000 iterations = 100 steps


def test_the_drivers_command_line_parses():
    import bench
    a = bench.make_parser().parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5) and a.partition == "temps"
    assert bench.TSKIP == 100 and bench.HBM_PEAK_GBS == 8000.0

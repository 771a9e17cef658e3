import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """Token-parity statistics of the session (tests/_cases.check_token_parity): per test, the number of draws, of
    disagreements and the largest |cdf - u| at one -- the evidence behind the tie tolerance."""
    try:
        import json
        import _cases
        if _cases.TIE_STATS:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            tot = {"draws": sum(v["draws"] for v in _cases.TIE_STATS.values()),
                   "disagreements": sum(v["disagreements"] for v in _cases.TIE_STATS.values()),
                   "max_tie_gap": max(v["max_tie_gap"] for v in _cases.TIE_STATS.values())}
            json.dump({"total": tot, "tests": _cases.TIE_STATS}, open(os.path.join(out, "tie_stats.json"), "w"), indent=1)
    except Exception as e:  # statistics only: never fail a session over them
        print("tie stats not written: %r" % (e,))

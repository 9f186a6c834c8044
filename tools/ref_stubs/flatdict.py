"""Stand-in for the `flatdict` package: re-exports this repo's own FlatterDict subset so the
reference SpectrumModel can be driven as an oracle (tools/gen_golden.py only)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from starfish_amd._flatdict import FlatterDict  # noqa: E402,F401

"""CPU-test launcher for bench.py's N>1 plumbing: swaps the GPU context (ecloop_amd.capi.Device) for tests/fake_device.py
(the oracle behind the same surface) and then runs bench.py's main().  bench.py itself knows nothing about this; the line
it prints is relabelled so that nobody can mistake it for a measurement."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import fake_device  # noqa: E402
from ecloop_amd import capi, engine  # noqa: E402

capi.Device = engine.Device = fake_device.FakeDevice
import bench  # noqa: E402

bench.visible_gpus = lambda: 1 << 20
bench.DATA = "synthetic (TEST STAND-IN for the device: not a measurement)"
bench.main()

"""ecloop_amd — MI355X-native drop-in for the hot path of vladkens/ecloop (`add`/`mul`/`rnd` key search).

Layout: csrc/ (HIP kernels + the C ABI of include/ecloop_hip.h), capi.py (ctypes binding of that ABI),
engine.py (host-side mirror of the reference's command drivers), host/ (the C command-line program)."""
from .capi import ADDR33, ADDR65, ENDO, Device, EclError  # noqa: F401

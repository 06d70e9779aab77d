"""The two CPU stand-ins for libmonodetr_amd.so used by the CPU tests of GPU-pending kernels:
  "host": tests/native_host.py  -- g++ build of plain loops around the shared *_math.h arithmetic;
  "emul": tests/native_emul.py  -- the real .hip kernels, launchers and capi.hip on the HIP-on-CPU shim."""
import native_emul
import native_host

BACKENDS = ["host", "emul"]


def get(name):
    return native_host.lib() if name == "host" else native_emul.lib()

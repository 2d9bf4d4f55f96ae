"""geometric_adv_amd -- MI355X-native hot path of itailang/geometric_adv (attack loop only).

The package holds the host-side mirror of the reference interface for that path; all arithmetic
runs in geometric_adv_amd/lib/libgeoadv.so (hand-written gfx950 HIP kernels behind the C ABI of
include/geoadv.h).  Importing the package does not need a GPU; calling any op does, and raises if
the library is not built."""
__version__ = "0.1.0"

"""dpu_olap_amd -- MI355X (gfx950) implementation of upmem/dpu_olap's operator hot path:
filter, take, sum, radix partition and radix-partitioned PK-FK hash join over Arrow
fixed-width integer columns.  C ABI: include/olap_gpu.h (csrc/*.hip -> lib/libolap_gpu.so);
this package holds the ctypes binding and the Python mirror of the reference's operator
classes.  There is no CPU fallback."""
from .capi import Context, OlapError, load_library  # noqa: F401
from .operators import (  # noqa: F401
    Device, DeviceArray, FilterGpu, JoinGpu, PartitionGpu, SumGpu, TakeGpu, FILTER_THRESHOLD,
)

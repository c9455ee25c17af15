"""bds_amd -- MI355X-native acquisition / tracking correlators for BDS-3 B1C and B2a.

Host-side mirror of the reference's MATLAB call surface
(lyf8118/BDS-3-B1C-B2a-SDR-receiver):

    settings   = init_settings_b1c() / init_settings_b2a()      initSettings.m
    acqResults = acquisition(longSignal, settings)              acquisition.m / GPU_acquisition.m
    channel    = pre_run(acqResults, settings)                  include/preRun.m
    trackResults, channel = tracking(fid, channel, settings)    tracking.m / NB_tracking.m / WB_tracking.m

All numeric work happens in ``libbds_mi355x.so`` (HIP kernels for gfx950) through
the C ABI of ``include/bds_mi355x.h``; there is no CPU fallback.
"""
from .settings import Settings, init_settings_b1c, init_settings_b2a  # noqa: F401
from .acquisition import acquisition, GPU_acquisition, AcqResults, get_context, release_context  # noqa: F401
from .tracking import tracking, NB_tracking, WB_tracking, pre_run, acquire_track, TrackResults  # noqa: F401
from .framesync import frame_sync, unpack_cplx  # noqa: F401
from .distributed import shard_prns, shard_joint, sharded_acquisition, sharded_acquisition_joint  # noqa: F401
from . import native, synth  # noqa: F401

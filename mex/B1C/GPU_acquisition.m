function acqResults = GPU_acquisition(longSignal, settings)
% Drop-in replacement of BDS-3_B1C/GPU_acquisition.m: same signature and acqResults fields; the
% search runs on the MI355X through bds_mex -> libbds_mi355x.so.
acqResults = bds_acquire_common(longSignal, settings, 1);
end

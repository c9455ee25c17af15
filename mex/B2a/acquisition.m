function acqResults = acquisition(longSignal, settings)
% Drop-in replacement of BDS-3_B2a/acquisition.m (same signature / result fields).
acqResults = bds_acquire_common(longSignal, settings, 2);
end

function [trackResults, channel] = tracking(fid, channel, settings)
% Drop-in replacement of BDS-3_B2a/tracking.m (same signature / result fields).
[trackResults, channel] = bds_track_common(fid, channel, settings, 2);
end

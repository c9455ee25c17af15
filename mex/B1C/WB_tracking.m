function [trackResults, channel] = WB_tracking(fid, channel, settings)
% Drop-in replacement of BDS-3_B1C/WB_tracking.m (pilotTRKflag == 2 selects the BOC(6,1) branch,
% WB_tracking.m:78).
if settings.pilotTRKflag ~= 2, settings.pilotTRKflag = 0; end
[trackResults, channel] = bds_track_common(fid, channel, settings, 1);
end

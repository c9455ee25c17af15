function [trackResults, channel] = NB_tracking(fid, channel, settings)
% Drop-in replacement of BDS-3_B1C/NB_tracking.m.  NB_tracking.m only tests pilotTRKflag == 1
% (NB_tracking.m:78); a struct with flag 2 tracks data-only there, so pass 0 in that case.
if settings.pilotTRKflag == 2, settings.pilotTRKflag = 0; end
[trackResults, channel] = bds_track_common(fid, channel, settings, 1);
end

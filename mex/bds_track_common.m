function [trackResults, channel] = bds_track_common(fid, channel, settings, signal)
% Shared body of the drop-in tracking wrappers.  The reference seeks absolutely from
% 'bof' for every channel (tracking.m:151-153), so only the file NAME is needed.
path = fopen(fid);
out = bds_mex('track', path, channel, settings, signal);
nCh = numel(channel);
skip = {'completed', 'status'};
names = setdiff(fieldnames(out), skip, 'stable');
sigName = 'B1C_CNo';
if signal == 2, sigName = 'B2a_CNo'; end
trackResults = struct([]);
for ch = 1:nCh
    r.status = char(out.status(ch));
    for k = 1:numel(names)
        n = names{k};
        dst = n;
        if strcmp(n, 'SigCNo'), dst = sigName; end
        r.(dst) = out.(n)(:, ch).';          % 1 x nEpochs rows, as tracking.m:51-93 allocates
    end
    if channel(ch).PRN ~= 0
        r.PRN = channel(ch).PRN;            % tracking.m:144
    else
        r.PRN = [];
    end
    trackResults = [trackResults r]; %#ok<AGROW>
end
if any(out.completed(:).' < size(out.I_P, 1) & [channel.PRN] ~= 0)
    disp('Not able to read the specified number of samples  for tracking, exiting!')  % tracking.m:251
end
end

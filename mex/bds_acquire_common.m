function acqResults = bds_acquire_common(longSignal, settings, signal)
% Shared body of the drop-in acquisition wrappers: converts the MATLAB arguments,
% calls the MEX gateway (bds_mex -> libbds_mi355x.so) and prints the reference's
% "(19 20 . )" line (acquisition.m:167,259,360,366).
%   signal: 1 = B1C, 2 = B2a
iq = ~isreal(longSignal);
if iq
    % fileType 2: longSignal = data(1:2:end) + 1i*data(2:2:end) (postProcessing.m:92-96);
    % hand the int8 pairs back interleaved
    pairs = [real(longSignal(:)).'; imag(longSignal(:)).'];
    longSignal = pairs(:).';
end
if any(longSignal ~= round(longSignal)) || any(longSignal > 127) || any(longSignal < -128)
    error('bds:arg', 'longSignal must hold int8 values (fread(...,''schar''))');
end
[carrFreq, codePhase, peakMetric, detected] = bds_mex('acquire', int8(longSignal), settings, signal, iq);
acqResults.carrFreq   = carrFreq;
acqResults.codePhase  = codePhase;
acqResults.peakMetric = peakMetric;
fprintf('(');
for PRN = settings.acqSatelliteList
    if detected(PRN), fprintf('%02d ', PRN); else, fprintf('. '); end
end
fprintf(')\n');
end

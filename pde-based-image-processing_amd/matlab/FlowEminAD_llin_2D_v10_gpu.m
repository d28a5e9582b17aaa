function [U V] = FlowEminAD_llin_2D_v10_gpu(Iin, channels, fstTerm, sndTerm, varargin)
%[U V] = FlowEminAD_llin_2D_v10_gpu(Iin, channels, fstTerm, sndTerm, varargin)
%
%Same call as FlowEminAD_llin_2D_v10 (matlab/optical_flow/FlowEminAD_llin_2D_v10.m of the toolbox); the whole coarse-to-fine
%run happens on the GPU in one MEX call (mex/FlowEminAD_llin_2D_v10_gpu.c -> libpdeip.so pdeip_flow_ad_llin).
%NOT RUN IN THIS REPOSITORY (no MATLAB in its build image); the MEX entry is tested through a mock MEX runtime.
param.alpha = 0; param.omega = 0; param.gammaS = 0; param.firstLoop = 0; param.secondLoop = 0; param.iter = 0;
param.b1 = 0; param.b2 = 0; param.scl_factor = 0; param.solver = 0; param.scales = 0; param.quantile = 0;	%0 = the driver's default
param.diffusion = 'image';
param.Us = []; param.Vs = [];
param = setParameters(param, varargin{:});
codes = struct('NONE', 0, 'RGB', 1, 'GRAD', 2, 'GRADMAG', 3);
pv = single([param.alpha param.omega param.gammaS param.firstLoop param.secondLoop param.iter ...
             param.b1 param.b2 param.scl_factor param.solver param.scales param.quantile strcmpi(param.diffusion, 'flow')]);
[U V] = FlowEminAD_llin_2D_v10_mex(single(Iin), single(channels), single(codes.(upper(fstTerm))), single(codes.(upper(sndTerm))), ...
                                   pv, double(param.Us), double(param.Vs));

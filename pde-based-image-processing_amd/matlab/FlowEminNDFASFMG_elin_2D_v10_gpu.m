function [U V] = FlowEminNDFASFMG_elin_2D_v10_gpu(Iin, channels, varargin)
%[U V] = FlowEminNDFASFMG_elin_2D_v10_gpu(Iin, channels, varargin)
%
%Same call as FlowEminNDFASFMG_elin_2D_v10 (matlab/optical_flow/FlowEminNDFASFMG_elin_2D_v10.m of the toolbox); the whole full-multigrid
%run happens on the GPU in one MEX call (mex/FlowEminNDFASFMG_elin_2D_v10_gpu.c -> libpdeip.so pdeip_flow_fas_fmg_elin).
%NOT RUN IN THIS REPOSITORY (no MATLAB in its build image); the MEX entry is tested through a mock MEX runtime.
param.alpha = 0; param.omega = 0; param.firstLoop = 0; param.iter = 0; param.b1 = 0; param.b2 = 0; param.scl_factor = 0;
param.solver = 0; param.cycle_index = 0; param.scales = 0;	%0 = the driver's default
param = setParameters(param, varargin{:});
pv = single([param.alpha param.omega param.firstLoop param.iter param.b1 param.b2 param.scl_factor param.solver param.cycle_index param.scales]);
[U V] = FlowEminNDFASFMG_elin_2D_v10_mex(single(Iin), single(channels), pv);

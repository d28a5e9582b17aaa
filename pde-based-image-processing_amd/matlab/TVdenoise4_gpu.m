function Iout = TVdenoise4_gpu(I_in, varargin)
%Iout = TVdenoise4_gpu(I_in, varargin)
%
%Same call as TVdenoise4 (matlab/denoising/TVdenoise4.m of the toolbox); the whole run happens on the GPU in one MEX call
%(mex/TVdenoise4_gpu.c -> libpdeip.so pdeip_tvdenoise4).
%NOT RUN IN THIS REPOSITORY (no MATLAB in its build image); the MEX entry is tested through a mock MEX runtime.
param.alpha = 0; param.omega = 0; param.outer_iter = 0; param.inner_iter = 0; param.solver = 0; param.scl = 0; param.scl_factor = 0;	%0 = the driver's default
param = setParameters(param, varargin{:});
pv = single([param.alpha param.omega param.outer_iter param.inner_iter param.solver param.scl param.scl_factor]);
Iout = TVdenoise4_mex(single(I_in), pv);

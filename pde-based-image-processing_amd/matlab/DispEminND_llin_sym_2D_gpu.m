function U = DispEminND_llin_sym_2D_gpu(Il, Ir, varargin)
%U = DispEminND_llin_sym_2D_gpu(Il, Ir, varargin)
%
%Same call as DispEminND_llin_sym_2D (matlab/disparity/DispEminND_llin_sym_2D.m of the toolbox: fstTerm / sndTerm are accepted and
%ignored as there); the whole coarse-to-fine run happens on the GPU in one MEX call (mex/DispEminND_llin_sym_2D_gpu.c ->
%libpdeip.so pdeip_disp_nd_llin_sym).
%NOT RUN IN THIS REPOSITORY (no MATLAB in its build image); the MEX entry is tested through a mock MEX runtime.
if numel(varargin) >= 2 && ischar(varargin{1}) && any(strcmpi(varargin{1}, {'rgb','grad'})) && any(strcmpi(varargin{2}, {'none','rgb','gradmag'}))
	varargin = varargin(3:end);	%runme.m:28 passes 'grad', 'gradmag'
end
param.alpha = 0; param.beta = 0; param.omega = 0; param.firstLoop = 0; param.secondLoop = 0; param.iter = 0;
param.b1 = 0; param.b2 = 0; param.scl_factor = 0; param.solver = 0;	%0 = the driver's default
param = setParameters(param, varargin{:});
pv = single([param.alpha param.beta param.omega param.firstLoop param.secondLoop param.iter param.b1 param.b2 param.scl_factor param.solver]);
U = DispEminND_llin_sym_2D_mex(single(Il), single(Ir), pv);

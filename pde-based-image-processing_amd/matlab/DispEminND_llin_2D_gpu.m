function U = DispEminND_llin_2D_gpu(Il, Ir, fstTerm, sndTerm, varargin)
%U = DispEminND_llin_2D_gpu(Il, Ir, fstTerm, sndTerm, varargin)
%
%Same call as DispEminND_llin_2D (matlab/disparity/DispEminND_llin_2D.m of the toolbox); the whole coarse-to-fine
%run happens on the GPU in one MEX call (mex/DispEminND_llin_2D_gpu.c -> libpdeip.so pdeip_disp_nd_llin).
%NOT RUN IN THIS REPOSITORY (no MATLAB in its build image); the MEX entry is tested through a mock MEX runtime.
param.alpha = 0; param.omega = 0; param.gammaS = 0; param.firstLoop = 0; param.secondLoop = 0; param.iter = 0;
param.b1 = 0; param.b2 = 0; param.scl_factor = 0; param.solver = 0; param.scales = 0;	%0 = the driver's default
param.Us = [];
param = setParameters(param, varargin{:});
codes = struct('NONE', 0, 'RGB', 1, 'GRAD', 2, 'GRADMAG', 3);
pv = single([param.alpha param.omega param.gammaS param.firstLoop param.secondLoop param.iter ...
             param.b1 param.b2 param.scl_factor param.solver param.scales]);
U = DispEminND_llin_2D_mex(single(Il), single(Ir), single(codes.(upper(fstTerm))), single(codes.(upper(sndTerm))), ...
                           pv, double(param.Us));

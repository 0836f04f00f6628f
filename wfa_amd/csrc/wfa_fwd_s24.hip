// wfa_fwd_s24.hip -- the sub-wave forward kernels for penalty shape x/g : (o+e)/g = 2 : 4 (wfa_fwd.hpp)
#define WFA_SHAPE_DX 2
#define WFA_SHAPE_DOE 4
#define WFA_SHAPE_TAG s24
#include "wfa_fwd_shape.inc"

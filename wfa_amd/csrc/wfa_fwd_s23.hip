// wfa_fwd_s23.hip -- the sub-wave forward kernels for penalty shape x/g : (o+e)/g = 2 : 3 (wfa_fwd.hpp)
#define WFA_SHAPE_DX 2
#define WFA_SHAPE_DOE 3
#define WFA_SHAPE_TAG s23
#include "wfa_fwd_shape.inc"

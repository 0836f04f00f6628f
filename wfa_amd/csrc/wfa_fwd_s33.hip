// wfa_fwd_s33.hip -- the sub-wave forward kernels for penalty shape x/g : (o+e)/g = 3 : 3 (wfa_fwd.hpp)
#define WFA_SHAPE_DX 3
#define WFA_SHAPE_DOE 3
#define WFA_SHAPE_TAG s33
#include "wfa_fwd_shape.inc"

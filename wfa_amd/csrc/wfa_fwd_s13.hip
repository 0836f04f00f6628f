// wfa_fwd_s13.hip -- the sub-wave forward kernels for penalty shape x/g : (o+e)/g = 1 : 3 (wfa_fwd.hpp)
#define WFA_SHAPE_DX 1
#define WFA_SHAPE_DOE 3
#define WFA_SHAPE_TAG s13
#include "wfa_fwd_shape.inc"

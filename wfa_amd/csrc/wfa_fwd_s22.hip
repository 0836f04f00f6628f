// wfa_fwd_s22.hip -- the sub-wave forward kernels for penalty shape x/g : (o+e)/g = 2 : 2 (wfa_fwd.hpp)
#define WFA_SHAPE_DX 2
#define WFA_SHAPE_DOE 2
#define WFA_SHAPE_TAG s22
#include "wfa_fwd_shape.inc"

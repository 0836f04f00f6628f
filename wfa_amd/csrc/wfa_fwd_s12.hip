// wfa_fwd_s12.hip -- the sub-wave forward kernels for penalty shape x/g : (o+e)/g = 1 : 2 (wfa_fwd.hpp)
#define WFA_SHAPE_DX 1
#define WFA_SHAPE_DOE 2
#define WFA_SHAPE_TAG s12
#include "wfa_fwd_shape.inc"

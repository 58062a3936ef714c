// Last include of every .hip file: closes the attribute region opened at the end of wm_common.h.
#ifdef WM_PK_GUARD
#pragma clang attribute pop
#undef WM_PK_GUARD
#endif

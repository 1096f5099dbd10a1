/*
 * dsputils.h -- symbol visibility macro of the library's public headers
 * (ref: include/dsputils.h:1-12).
 *
 * The reference defines EXPORT as default visibility on Linux / dllexport on WIN32 only when it is
 * itself being built (-DDLL_EXPORT) and as nothing for callers.  This library is ELF-only (gfx950
 * hosts are Linux); default visibility is harmless on a caller's declarations, so one spelling
 * serves both sides.  DLL_EXPORT is accepted and changes nothing.
 */
#ifndef DSPUTILS_H
#define DSPUTILS_H

#ifndef EXPORT
#  if defined(__GNUC__) || defined(__clang__)
#    define EXPORT __attribute__((visibility("default")))
#  else
#    define EXPORT
#  endif
#endif

#endif /* DSPUTILS_H */

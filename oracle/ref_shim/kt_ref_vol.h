/* internal.h:243 fixes the volume side at compile time (#define VOL 512; enum { VOLUME_X = VOL, ... }).  The recipe
 * includes this header right after the reference's own headers in tsdf_volume.cu, ray_caster.cu and extract.cu so that
 * one libkt_ref.so serves every N the parity tests use (ktref_set_volume_side); with N = 512 the code is the stock build. */
#ifndef KT_REF_VOL_H_
#define KT_REF_VOL_H_
extern int ktref_volume_side;
#define VOLUME_X ktref_volume_side
#define VOLUME_Y ktref_volume_side
#define VOLUME_Z ktref_volume_side
#endif

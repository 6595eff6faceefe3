// sd_fast_fl_u16s.hip -- the u16 kernels of sd_fast_fl_u16.hip with ONE floor level for every row and the floor inside the slot
// loop (not in place): scorings with a negative table value (a mismatch that costs more than a deletion plus an insertion),
// where raising a slot's "keep" operand to the start term would not be harmless (sd_fast_fill.hpp: FLS).  P = 30..40.
#define SD_FL_CF CF_U16
#define SD_FL_STEP 0
#define SD_FL_ENTRY launch_fast_fill_fl_u16s
#define SD_FL_ENTRY_LONG launch_fast_fill_fl_long_u16s
#define SD_FL_TAKES(plan) ((plan).u16)
#include "sd_fast_fl.hip"

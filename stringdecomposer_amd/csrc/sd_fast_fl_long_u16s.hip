// sd_fast_fl_long_u16s.hip -- the one-level u16 kernels for P = 42..64 (see sd_fast_fl_u16s.hip).
#define SD_FL_CF CF_U16
#define SD_FL_STEP 0
#define SD_FL_ENTRY_LONG launch_fast_fill_fl_long_u16s
#define SD_FL_TAKES(plan) ((plan).u16)
#include "sd_fast_fl_long.hip"

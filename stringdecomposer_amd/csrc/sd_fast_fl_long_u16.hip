// sd_fast_fl_long_u16.hip -- the kernels of sd_fast_fl_long.hip (P = 42..64) for the biased-u16 cell format.
#define SD_FL_CF CF_U16
#define SD_FL_STEP 4
#define SD_FL_ENTRY_LONG launch_fast_fill_fl_long_u16
#define SD_FL_TAKES(plan) ((plan).u16 && (plan).table_nonneg && !getenv("SD_FILL_ONE_LEVEL"))
#include "sd_fast_fl_long.hip"

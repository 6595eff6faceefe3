// sd_fast_fl_u16.hip -- the kernels of sd_fast_fl.hip (start-term maximum in the first FL slots only, P = 30..40) for the
// biased-u16 cell format (CellOps<CF_U16>, sd_fast_dev.hpp): the narrow layout's default since round 6.
#define SD_FL_CF CF_U16
#define SD_FL_STEP 4
#define SD_FL_ENTRY launch_fast_fill_fl_u16
#define SD_FL_ENTRY_LONG launch_fast_fill_fl_long_u16
#define SD_FL_TAKES(plan) ((plan).u16 && (plan).table_nonneg && !getenv("SD_FILL_ONE_LEVEL"))
#include "sd_fast_fl.hip"

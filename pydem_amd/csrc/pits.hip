// pits.hip -- pit -> drain assignment (placeholder until the device implementation lands)
#include "internal.h"

int stage_pits(pydem_tile *t, const pydem_options *opt)
{
    (void)t; (void)opt;
    pydem_set_error("drain_pits=True is not implemented on the device yet");
    return -4;
}

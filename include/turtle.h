/*
 * turtle.h -- drop-in name for callers written against the reference's
 * public header: `#include "turtle.h"` with -I<this directory> and linking
 * -lturtle_amd gives the stepper path of the reference API on an MI355X.
 * All declarations live in turtle_amd.h.
 */
#ifndef TURTLE_H
#define TURTLE_H
#include "turtle_amd.h"
#endif

// The float chain with int16 samples in (sa_process_f32_i16): chain_f32.hip compiled a second time with the int16
// stage-in.  Everything behind the stage-in -- cascade, FFT, split step, outputs -- is the same source.
#undef SA_STAMPS                 // the diagnostic stamps belong to the float32 translation unit
#define SA_F32_INPUT_I16 1
#include "chain_f32.hip"

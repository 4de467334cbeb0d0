#include <hip/hip_runtime.h>
#include <cstdio>
int main(){int k=0; hipError_t e=hipDeviceGetAttribute(&k, hipDeviceAttributeWallClockRate, 0); printf("hipDeviceAttributeWallClockRate rc %d value %d kHz\n",(int)e,k); return 0;}

#pragma once
// Same include path shape as the reference (`#include <library/msufsort.h>`,
// reference src/library/msufsort.h:1-4, used at src/executable/msufsort/main.cpp:8).
#include "./msufsort/msufsort.h"

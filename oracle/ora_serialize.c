/*
 * oracle/ora_serialize.c — CPU restatement of the sink-side marshalling
 * (SURVEY.md §8a rows a20, a21).  TEST INFRASTRUCTURE ONLY (see ora.h).
 */
#include <stdlib.h>
#include <string.h>
#include "ora.h"

char *ora_serialize(int format, const ora_batch *b, uint64_t *len) {
  (void)format; (void)b;
  *len = 0;
  return NULL; /* filled in by the serializer milestone */
}

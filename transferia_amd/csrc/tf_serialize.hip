// tf_serialize.hip — sink-side marshalling from device columns (SURVEY.md §8a
// a20/a21).  Placeholder until the serializer milestone: fails loudly.
#include "tf_common.hpp"

extern "C" int tfgpu_serialize(int format, const tfgpu_dbatch *b, tfgpu_dbuf **out) {
  (void)format; (void)b; (void)out;
  return tf::fail(TFGPU_ERR_UNSUPPORTED, "tfgpu_serialize: not implemented yet");
}

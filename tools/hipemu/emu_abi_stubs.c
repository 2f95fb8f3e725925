/* tools/hipemu/emu_abi_stubs.c — TEST INFRASTRUCTURE ONLY.  The entry points whose kernels are not part of the hipemu
 * build (they use LDS / barriers / wave intrinsics) answer TFGPU_ERR_UNSUPPORTED (3) so the ctypes binding still loads. */
#define STUB(name) int name() { return 3; }
STUB(tfgpu_plan_create) STUB(tfgpu_plan_description) STUB(tfgpu_plan_suitable) STUB(tfgpu_plan_result_schema) STUB(tfgpu_apply)
STUB(tfgpu_collapse) STUB(tfgpu_keys_changed) STUB(tfgpu_partition) STUB(tfgpu_csv_parse) 
STUB(tfgpu_registry_count)
void tfgpu_plan_destroy(void *p) { (void)p; }
const char *tfgpu_plan_type(const void *p) { (void)p; return 0; }
const char *tfgpu_registry_name(int i) { (void)i; return 0; }
void tfgpu_schema_free(void *p) { (void)p; }
void tfgpu_csv_options_default(void *p) { (void)p; }

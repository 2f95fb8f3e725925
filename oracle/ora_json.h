/* oracle/ora_json.h — minimal JSON tree reader for transformer configs.
 * TEST INFRASTRUCTURE ONLY (see ora.h). */
#ifndef ORA_JSON_H
#define ORA_JSON_H
#include <stddef.h>
typedef enum { JN_NULL, JN_BOOL, JN_NUM, JN_STR, JN_ARR, JN_OBJ } jn_type;
typedef struct jnode {
  jn_type type;
  int b;
  double num;
  char *str; size_t slen;     /* JN_STR (unescaped) / JN_NUM (raw text) */
  int n;                      /* children */
  struct jnode **kids;
  char **keys;                /* JN_OBJ */
} jnode;
jnode *jn_parse(const char *text, char *err, size_t errcap);
void jn_free(jnode *n);
const jnode *jn_get(const jnode *obj, const char *key);
const char *jn_str(const jnode *n, const char *dflt);
int jn_bool(const jnode *n, int dflt);
#endif

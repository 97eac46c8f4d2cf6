// In-memory stand-in for liblmdb (third-party, absent here), just enough for the reference's CustomDataLayer to be compiled in
// place and run by the pin harness: an "environment" is a sorted key -> value map registered under a source name by
// oracle/ref_shim.cpp.  Keys compare bytewise like LMDB's default comparator.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstddef>
#include <map>
#include <string>

struct MDB_val { size_t mv_size; void* mv_data; };
struct MDB_stat { size_t ms_entries; };
typedef std::map<std::string, std::string> fn2_fake_db;
struct MDB_env { const fn2_fake_db* db; };
struct MDB_txn { MDB_env* env; };
struct MDB_cursor { MDB_env* env; fn2_fake_db::const_iterator it; };
typedef unsigned int MDB_dbi;
enum MDB_cursor_op { MDB_FIRST, MDB_GET_CURRENT, MDB_NEXT, MDB_SET_RANGE };
#define MDB_SUCCESS 0
#define MDB_NOTFOUND (-30798)
#define MDB_RDONLY 0x20000
#define MDB_NOTLS 0x200000

inline std::map<std::string, fn2_fake_db>& fn2_fake_lmdb_sources() { static std::map<std::string, fn2_fake_db> s; return s; }

inline int mdb_env_create(MDB_env** env) { *env = new MDB_env{nullptr}; return MDB_SUCCESS; }
inline int mdb_env_set_mapsize(MDB_env*, size_t) { return MDB_SUCCESS; }
inline int mdb_env_open(MDB_env* env, const char* path, unsigned int, int) {
  auto it = fn2_fake_lmdb_sources().find(path);
  if (it == fn2_fake_lmdb_sources().end()) return 2;   // ENOENT
  env->db = &it->second;
  return MDB_SUCCESS;
}
inline int mdb_reader_check(MDB_env*, int* dead) { *dead = 0; return MDB_SUCCESS; }
inline int mdb_txn_begin(MDB_env* env, MDB_txn*, unsigned int, MDB_txn** txn) { *txn = new MDB_txn{env}; return MDB_SUCCESS; }
inline int mdb_open(MDB_txn*, const char*, unsigned int, MDB_dbi* dbi) { *dbi = 1; return MDB_SUCCESS; }
inline int mdb_cursor_open(MDB_txn* txn, MDB_dbi, MDB_cursor** cur) { *cur = new MDB_cursor{txn->env, txn->env->db->begin()}; return MDB_SUCCESS; }
inline int mdb_stat(MDB_txn* txn, MDB_dbi, MDB_stat* st) { st->ms_entries = txn->env->db->size(); return MDB_SUCCESS; }
inline int mdb_cursor_get(MDB_cursor* cur, MDB_val* key, MDB_val* val, MDB_cursor_op op) {
  const fn2_fake_db& db = *cur->env->db;
  if (op == MDB_FIRST) cur->it = db.begin();
  else if (op == MDB_NEXT) { if (cur->it != db.end()) ++cur->it; }
  else if (op == MDB_SET_RANGE) cur->it = db.lower_bound(std::string(static_cast<const char*>(key->mv_data), key->mv_size));
  if (cur->it == db.end()) return MDB_NOTFOUND;
  key->mv_size = cur->it->first.size();
  key->mv_data = const_cast<char*>(cur->it->first.data());
  val->mv_size = cur->it->second.size();
  val->mv_data = const_cast<char*>(cur->it->second.data());
  return MDB_SUCCESS;
}
inline void mdb_cursor_close(MDB_cursor* cur) { delete cur; }
inline void mdb_close(MDB_env*, MDB_dbi) {}
inline void mdb_txn_abort(MDB_txn* txn) { delete txn; }
inline void mdb_env_close(MDB_env* env) { delete env; }

// TEST INFRASTRUCTURE ONLY.  C entry point around the reference's own OBJ loader, compiled from the
// reference sources where they lie (/root/reference/external/src/objUtil/{obj,objloader}.cpp) by
// `make -C oracle ref` into oracle/_ref/libobjref.so.  It repeats Scene::loadObjFile + Scene::objToMesh
// (reference src/world/scene.cpp:26-33, :115-133): objLoader -> buildVBOs -> VBO/TBO/bbox picks.
// Used to pin ora_mesh_load_obj (and the product's svoslam_mesh_load_obj through it) against the
// reference itself; nothing else of the reference compiles here.
#include <objUtil/obj.h>
#include <objUtil/objloader.h>

#include <cstdlib>
#include <cstring>
#include <string>

extern "C" {

// Returns the triangle count (vbosize / 9) or -1; *vbo / *tbo are malloc'ed copies (ref_obj_free).
int ref_obj_load(const char* path, float** vbo, float** tbo, int* tbosize, float bbox0[3], float bbox1[3]) {
  FILE* probe = fopen(path, "rb");
  if (!probe) return -1;
  fclose(probe);
  obj* o = new obj();
  objLoader loader(std::string(path), o);
  o->buildVBOs();
  const int vn = o->getVBOsize(), tn = o->getTBOsize();
  *vbo = static_cast<float*>(malloc(sizeof(float) * (vn > 0 ? vn : 1)));
  memcpy(*vbo, o->getVBO(), sizeof(float) * vn);
  *tbo = nullptr;
  if (tn > 0) {
    *tbo = static_cast<float*>(malloc(sizeof(float) * tn));
    memcpy(*tbo, o->getTBO(), sizeof(float) * tn);
  }
  *tbosize = tn;
  const float* bb = o->getBoundingBox();
  bbox0[0] = bb[0]; bbox0[1] = bb[1]; bbox0[2] = bb[18];
  bbox1[0] = bb[8]; bbox1[1] = bb[5]; bbox1[2] = bb[2];
  delete o;
  return vn / 9;
}

void ref_obj_free(void* p) { free(p); }

}  // extern "C"

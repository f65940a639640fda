// Test-infrastructure stub for MarchingCubeCpp's MC.h (not on disk): SDF-plugin
// meshes only; produces an empty mesh.
#ifndef ORACLE_STUB_MC_H_
#define ORACLE_STUB_MC_H_
#include <vector>
namespace MC {
typedef float MC_FLOAT;
struct mcVec3f { MC_FLOAT x, y, z; };
struct mcMesh { std::vector<mcVec3f> vertices, normals; std::vector<unsigned int> indices; };
inline void marching_cube(MC_FLOAT*, int, int, int, mcMesh&) {}
}
#endif

// amg.hip -- placeholder until the device V-cycle lands (this file is replaced later in round 1).
#include "amg.hpp"

#include "solver.hpp"

namespace psolve {

struct AmgHierarchy::Impl {
    int nlevels = 0;
};

AmgHierarchy::AmgHierarchy() : impl(new Impl()) {}
AmgHierarchy::~AmgHierarchy() = default;

void AmgHierarchy::setup(Context &, const CsrDev &, const AmgParams &)
{
    throw Error(PSOLVE_HIP_EINVAL, "precond=amg is not built yet");
}

void AmgHierarchy::apply(Context &, const double *, double *)
{
    throw Error(PSOLVE_HIP_EINVAL, "precond=amg is not built yet");
}

int AmgHierarchy::levels() const { return impl->nlevels; }

} // namespace psolve

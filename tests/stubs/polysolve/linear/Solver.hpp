// Stand-in for the PolySolve interface headers, written for this repository's tests only.
//
// polysolve_amd/host/HIPSolver.hpp derives from polysolve::linear::Solver and speaks Eigen and
// nlohmann::json, none of which exist in this image.  This file declares JUST the surface the adapter
// touches -- a sparse-matrix class with Eigen's accessor names, a contiguous vector, a Ref that wraps a
// pointer, a small JSON value with nlohmann's member names, and the abstract Solver with the virtuals of
// the reference's Solver.hpp:90-131 -- so that the adapter can be compiled (g++ -fsyntax-only, CPU test) and
// driven against libpsolve_hip.so (GPU test, tests/adapter_driver.cpp).  It is not a copy of any upstream
// header and is never shipped: a PolySolve build uses its own headers.
#pragma once
#include <Eigen/Sparse>
#include <cstddef>
#include <iterator>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

#define POLYSOLVE_DELETE_MOVE_COPY(Base) \
    Base(Base &&) = delete;              \
    Base &operator=(Base &&) = delete;   \
    Base(const Base &) = delete;         \
    Base &operator=(const Base &) = delete;

namespace polysolve
{
    // (Types.hpp:11-15)
#ifdef POLYSOLVE_LARGE_INDEX
    typedef Eigen::SparseMatrix<double, Eigen::ColMajor, std::ptrdiff_t> StiffnessMatrix;
#else
    typedef Eigen::SparseMatrix<double, Eigen::ColMajor> StiffnessMatrix;
#endif

#ifdef PSOLVE_TEST_REAL_NLOHMANN
} // namespace polysolve
// tests/test_adapter.py, where the image has an nlohmann/json single header (this one: /opt/conda/include/json.hpp, 3.1.1,
// shipped with another package): the adapter is compiled and driven against the REAL json class, not the stand-in below
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wall"
#pragma GCC diagnostic ignored "-Wextra"
#pragma GCC diagnostic ignored "-Wdeprecated-declarations"
#include PSOLVE_TEST_REAL_NLOHMANN
#pragma GCC diagnostic pop
namespace polysolve
{
    using json = nlohmann::json;
#else
    // a JSON value with the nlohmann member names the adapter uses
    class json
    {
    public:
        using object_t = std::map<std::string, json>;
        using array_t = std::vector<json>;
        json() = default;
        json(bool b) : v_(b) {}
        template <typename T, typename = std::enable_if_t<std::is_arithmetic_v<T> && !std::is_same_v<T, bool>>>
        json(T number) : v_((double)number) {}
        json(const char *s) : v_(std::string(s)) {}
        json(std::string s) : v_(std::move(s)) {}
        json(std::initializer_list<std::pair<const std::string, json>> kv) : v_(object_t(kv)) {}
        static json array(std::initializer_list<json> items) { json j; j.v_ = array_t(items); return j; }

        bool is_null() const { return std::holds_alternative<std::monostate>(v_); }
        bool is_boolean() const { return std::holds_alternative<bool>(v_); }
        bool is_number() const { return std::holds_alternative<double>(v_); }
        bool is_string() const { return std::holds_alternative<std::string>(v_); }
        bool is_object() const { return std::holds_alternative<object_t>(v_); }
        bool is_array() const { return std::holds_alternative<array_t>(v_); }
        bool contains(const std::string &key) const { return is_object() && std::get<object_t>(v_).count(key) > 0; }

        const json &operator[](const std::string &key) const { return std::get<object_t>(v_).at(key); }
        json &operator[](const std::string &key)
        {
            if (is_null()) v_ = object_t();
            return std::get<object_t>(v_)[key];
        }
        template <typename T>
        T get() const
        {
            if constexpr (std::is_same_v<T, bool>) return std::get<bool>(v_);
            else if constexpr (std::is_same_v<T, std::string>) return std::get<std::string>(v_);
            else
            {
                if (!is_number()) throw std::runtime_error("json: not a number");
                return (T)std::get<double>(v_);
            }
        }
        operator std::string() const { return std::get<std::string>(v_); }

        size_t count(const std::string &key) const { return contains(key) ? 1 : 0; }
        // iteration as nlohmann's: over an object's members (it.key(), it.value()) or an array's elements (*it)
        class const_iterator
        {
        public:
            const_iterator(const json *j, size_t i) : j_(j), i_(i) {}
            bool operator!=(const const_iterator &o) const { return i_ != o.i_ || j_ != o.j_; }
            const_iterator &operator++() { ++i_; return *this; }
            const json &operator*() const { return value(); }
            std::string key() const
            {
                auto it = std::get<object_t>(j_->v_).begin();
                std::advance(it, (long)i_);
                return it->first;
            }
            const json &value() const
            {
                if (j_->is_array()) return std::get<array_t>(j_->v_)[i_];
                auto it = std::get<object_t>(j_->v_).begin();
                std::advance(it, (long)i_);
                return it->second;
            }

        private:
            const json *j_;
            size_t i_;
        };
        const_iterator begin() const { return const_iterator(this, 0); }
        const_iterator end() const
        {
            return const_iterator(this, is_object() ? std::get<object_t>(v_).size() : (is_array() ? std::get<array_t>(v_).size() : 0));
        }

        // items(): (key, value) pairs of an object, (index-as-string, value) pairs of an array
        std::vector<std::pair<std::string, json>> items() const
        {
            std::vector<std::pair<std::string, json>> out;
            if (is_object())
                for (const auto &kv : std::get<object_t>(v_)) out.emplace_back(kv.first, kv.second);
            else if (is_array())
            {
                size_t i = 0;
                for (const auto &e : std::get<array_t>(v_)) out.emplace_back(std::to_string(i++), e);
            }
            return out;
        }

    private:
        std::variant<std::monostate, bool, double, std::string, object_t, array_t> v_;
    };
#endif // PSOLVE_TEST_REAL_NLOHMANN

    namespace linear
    {
        class Solver
        {
        public:
            typedef Eigen::VectorXd VectorXd;
            template <typename T>
            using Ref = Eigen::Ref<T>;
            virtual ~Solver() = default;
            virtual void set_parameters(const json &) {}
            virtual void get_info(json &) const {}
            virtual void analyze_pattern(const StiffnessMatrix &, const int) {}
            virtual void factorize(const StiffnessMatrix &) {}
            virtual bool is_dense() const { return false; }
            virtual void set_block_size(int) {}
            virtual void set_is_nullspace(const VectorXd &) {}
            virtual void set_tolerance(const double) {}
            virtual void solve(const Ref<const VectorXd> b, Ref<VectorXd> x) = 0;
            virtual std::string name() const { return ""; }

        protected:
            Solver() = default;
        };
    } // namespace linear
} // namespace polysolve

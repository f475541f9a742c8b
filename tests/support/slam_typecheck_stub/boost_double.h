// NOT boost.  Declaration-only test double for `g++ -fsyntax-only` of this repository's ORBmatcher_hip.cc against the reference's
// unmodified headers (tests/test_adapter_typecheck.py).  The reference's serialize() member templates are never instantiated here.
#pragma once
namespace boost { namespace serialization {
class access;
template <class T> struct array_wrapper_double { };
template <class T> array_wrapper_double<T> make_array(T *, unsigned long);
template <class Base, class Derived> Base &base_object(Derived &d);
}  // namespace serialization
namespace archive { class binary_iarchive; class binary_oarchive; class text_iarchive; class text_oarchive; }
}  // namespace boost
#define BOOST_SERIALIZATION_SPLIT_MEMBER()
#define BOOST_SERIALIZATION_ASSUME_ABSTRACT(x)
#define BOOST_CLASS_EXPORT_KEY(x)
#define BOOST_CLASS_EXPORT_IMPLEMENT(x)

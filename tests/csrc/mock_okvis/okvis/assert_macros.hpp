// stand-in for okvis_util/include/okvis/assert_macros.hpp:54-119 (exception definition, throw / assert macros)
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>
#define OKVIS_DEFINE_EXCEPTION(exceptionName, exceptionParent)                       \
  class exceptionName : public exceptionParent {                                     \
   public:                                                                           \
    explicit exceptionName(const char* message) : exceptionParent(message) {}        \
    explicit exceptionName(std::string const& message) : exceptionParent(message) {} \
  };
#define OKVIS_THROW(exceptionType, message)          \
  {                                                  \
    std::stringstream okvis_assert_stringstream;     \
    okvis_assert_stringstream << message;            \
    throw exceptionType(okvis_assert_stringstream.str()); \
  }
#define OKVIS_ASSERT_TRUE(exceptionType, condition, message)                           \
  if (!(condition)) {                                                                  \
    std::stringstream okvis_assert_stringstream;                                       \
    okvis_assert_stringstream << "assert(" << #condition << ") failed: " << message;   \
    throw exceptionType(okvis_assert_stringstream.str());                              \
  }

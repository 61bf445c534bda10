#ifndef STD_MSGS_STRING_STUB_H
#define STD_MSGS_STRING_STUB_H
#include <memory>
#include <string>
namespace std_msgs {
struct String {
    std::string data;
    typedef std::shared_ptr<const String> ConstPtr;
};
}  // namespace std_msgs
#endif

// Vaa3D shell around libpnr_hip.so (see the header: needs Qt4 + the Vaa3D SDK, not built here).
// Same command-line contract as the reference plugin (Advantra_plugin.cpp:157-162, 274-337): functions
// "advantra_func" and "help"; input[0] = image file names, input[1] = the 11 positional parameters; a wrong
// parameter count prints the help and returns false; a range error prints the message and stops.  Everything
// from the loaded stack on is advantra::reconstruction_func (../advantra_host.cpp), i.e. the C ABI of
// include/pnr_hip.h -- this file only adapts Vaa3D's types.
#include "AdvantraHip_plugin.h"
#include "../advantra_host.h"
#include "basic_surf_objs.h"
#include "v3d_message.h"
#include <string>
#include <vector>

Q_EXPORT_PLUGIN2(Advantra, AdvantraHipPlugin);

QStringList AdvantraHipPlugin::menulist() const { return QStringList() << tr("about"); }

QStringList AdvantraHipPlugin::funclist() const { return QStringList() << tr("advantra_func") << tr("help"); }

void AdvantraHipPlugin::domenu(const QString &, V3DPluginCallback2 &, QWidget *)
{
    // the reference's menu entry opens a parameter dialog over the current image window (Advantra_plugin.cpp:176-272);
    // the GPU build is driven through dofunc only
    v3d_msg(tr("Advantra (HIP build): use  vaa3d -x Advantra -f advantra_func -i <inimg_file> -p <11 parameters>"));
}

bool AdvantraHipPlugin::dofunc(const QString &func_name, const V3DPluginArgList &input, V3DPluginArgList &, V3DPluginCallback2 &callback,
                      QWidget *)
{
    if (func_name == tr("help")) {
        advantra::print_help();
        return true;
    }
    if (func_name != tr("advantra_func")) return false;

    std::vector<char *> none;
    std::vector<char *> &infiles = input.size() >= 1 && input[0].p ? *(std::vector<char *> *)input[0].p : none;
    std::vector<char *> &paras_c = input.size() >= 2 && input[1].p ? *(std::vector<char *> *)input[1].p : none;
    if (infiles.empty()) {
        fprintf(stderr, "Need input image. \n");
        return false;
    }
    std::vector<std::string> paras(paras_c.begin(), paras_c.end());
    pnr_params prm;
    std::string err;
    const int pr = advantra::parse_params(paras, prm, err);
    if (pr == -1) {
        advantra::print_help();
        return false;
    }
    if (pr == -2) {
        v3d_msg(err.c_str(), 0);
        return true;
    }

    unsigned char *data1d = 0; // owned here, as in the reference's function-call mode
    V3DLONG in_sz[4] = {0, 0, 0, 0};
    int datatype = 0;
    if (!simple_loadimage_wrapper(callback, infiles[0], data1d, in_sz, datatype)) {
        fprintf(stderr, "Error happens in reading the subject file [%s]. Exit. \n", infiles[0]);
        return true;
    }
    if (datatype != 1) { // the tracer works on 8-bit stacks
        v3d_msg("Advantra needs an 8-bit image.", 0);
        delete[] data1d;
        return true;
    }
    // channel 1 = the first N*M*P bytes of data1d
    advantra::reconstruction_func(data1d, in_sz[0], in_sz[1], in_sz[2], infiles[0], paras, prm, /*device*/ 0);
    delete[] data1d;
    return true;
}
